#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do python -m pytest tests/test_gpu_stripes.py -m gpu -q -k "launch_shapes" 2>&1 | grep -E "AssertionError|passed|failed" | head -2 | cut -c1-200; done
