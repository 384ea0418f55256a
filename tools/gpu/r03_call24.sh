#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "adjoint_latent or own_kernel" 2>&1 | grep -E "^E  |assert|passed|failed" | head -40
