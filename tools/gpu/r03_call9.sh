#!/bin/bash
O=gpurun_out/r03c9; mkdir -p $O
for i in $(seq 1 12); do timeout 300 python -m pytest tests/test_gpu_stripes.py -q -m gpu -k "two_frames_in_flight" 2>&1 | grep -E "passed|failed|AssertionError: frame" ; done | tee $O/loop.txt
