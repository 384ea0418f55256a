#!/bin/bash
# Developer tool: builds libfvsrn variants with parts of the sample loop removed (FVSRN_ABL_* macros) HERE (no GPU
# needed), into fv-srn_amd/ablate/; tools/ablate_run.sh benches them on the GPU box.  The numbers bound what each part
# of the loop costs; the variants render wrong images by construction.
cd "$(dirname "$0")/.."
mkdir -p fv-srn_amd/ablate
for v in "$@"; do
  name=$(echo "$v" | tr -c 'A-Za-z0-9_\n' '_')
  make -C fv-srn_amd/csrc -j8 BUILD=build_abl_$name OUT=../ablate/libfvsrn_$name.so EXTRA="$v" 2>&1 | grep -E "error" -A3
  rm -rf fv-srn_amd/csrc/build_abl_$name
done
ls -la fv-srn_amd/ablate
