#!/bin/bash
# section split of the 64x6 + grid wave step (developer build with s_memtime marks)
export FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn__DFVSRN_PROF_SECTIONS.so
for mb in 0 1 2; do
  echo "== max_blocks_per_cu $mb"
  FVSRN_MAX_BLOCKS_PER_CU=$mb python tools/section_profile.py c64l6_grid16_1024x512 2>&1 | grep -v amdgpu.ids
done
echo "== waves per block"
for w in 1 2 4; do FVSRN_WAVES_PER_BLOCK=$w python tools/section_profile.py c64l6_grid16_1024x512 2>&1 | grep -v amdgpu.ids | head -1; done
unset FVSRN_LIBRARY
for w in 0 1 2 4; do FVSRN_WAVES_PER_BLOCK=$w bash tools/quick_bench.sh wpb$w --config c64l6_grid16_1024x512 --no-twin; done
