#!/bin/bash
# Developer tool (hazard bisect, r04), runs ON THE GPU BOX: links one library per variant object in fv-srn_amd/ablate/var/<tu>__<name>.o with the
# base objects in fv-srn_amd/ablate/base/ and runs run_cases.py on it.   usage: box_run.sh <tu> <family> [N] [repeats]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tu=$1; fam=$2; n=${3:-30}; reps=${4:-1}
base=""
for o in fv-srn_amd/ablate/base/*.o; do [ "$(basename $o .o)" = "$tu" ] || base="$base $o"; done
for v in fv-srn_amd/ablate/var/${tu}__*.o; do
  name=$(basename $v .o); name=${name#${tu}__}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libfvsrn_var.so $base $v || { echo "$name: link failed"; continue; }
  for r in $(seq $reps); do
    echo "== $name: $(FVSRN_LIBRARY=/tmp/libfvsrn_var.so timeout 300 python tools/dev/bisect/run_cases.py $fam $n 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-400)"
  done
done
