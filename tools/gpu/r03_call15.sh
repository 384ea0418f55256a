#!/bin/bash
O=gpurun_out/r03c15; mkdir -p $O
python tools/dev/ahead_repeat.py 2>&1 | grep -v amdgpu.ids | tee $O/ahead_repeat.txt
