#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== two frames at a time on two streams, scratch fence OFF"; FVSRN_NO_SCRATCH_FENCE=1 python tools/dev/stress_concurrent.py 100 2>&1 | grep -v amdgpu.ids | cut -c1-330
echo "== fence on"; python tools/dev/stress_concurrent.py 100 2>&1 | grep -v amdgpu.ids | cut -c1-330
