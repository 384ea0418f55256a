#!/bin/bash
# Developer tool (hazard bisect, r04), runs ON THE GPU BOX: run_aggr.py for the variant objects named on the command line (default 00base).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tu=kernels_cd4_stripe
base=""
for o in fv-srn_amd/ablate/base/*.o; do [ "$(basename $o .o)" = "$tu" ] || base="$base $o"; done
for name in ${@:-00base}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libfvsrn_var.so $base fv-srn_amd/ablate/var/${tu}__$name.o || { echo "$name: link failed"; continue; }
  echo "== $name"
  FVSRN_LIBRARY=/tmp/libfvsrn_var.so GPU_MAX_HW_QUEUES=8 timeout 600 python tools/dev/bisect/run_aggr.py ${AGGR_N:-20} ${AGGR_BLOCKS:-512} 2>&1 | grep -v amdgpu.ids
done
