#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in tapall taplo taphi; do for rep in 1 2; do echo "== $v ($rep)"; FVSRN_LIBRARY=$GRAFT_REPO_ROOT/fv-srn_amd/ablate/libfvsrn_$v.so python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "stripe_kernel" | cut -c1-120; done; done
echo "== shipped (1)"; python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "stripe_kernel" | cut -c1-120
echo "== shipped (2)"; python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "stripe_kernel" | cut -c1-120
