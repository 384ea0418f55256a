#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in tap0 tap1 tap2 tap4; do for rep in 1 2; do echo "== $v ($rep)"; FVSRN_LIBRARY=$GRAFT_REPO_ROOT/fv-srn_amd/ablate/libfvsrn_$v.so python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep "stripe_kernel" | awk '{print "   ", $0}' | cut -c1-110; done; done
