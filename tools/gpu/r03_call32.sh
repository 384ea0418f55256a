#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "autograd" -s 2>&1 | grep -E "passed|failed|adjoint evaluate|Error|assert" | head -20
