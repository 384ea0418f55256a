#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== shipped"; python tools/dev/determinism.py 20 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "64x3"
echo "== s_waitcnt vmcnt(0) between tiles"; FVSRN_LIBRARY=$GRAFT_REPO_ROOT/fv-srn_amd/ablate/libfvsrn_drain.so python tools/dev/determinism.py 20 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "64x3"
