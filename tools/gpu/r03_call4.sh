#!/bin/bash
# r03 call 4: rotating register-resident kernel with a latent chunk (per-ray rotation parked in LDS) vs the r02 kernel (direct features)
O=gpurun_out/r03c4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py tests/test_gpu_stripes.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for i in 1 2; do
  for c in c32l4_grid16_1024x512 c32l4_grid16r32_1024x512; do
    bash tools/quick_bench.sh rotate_lds --config $c
    FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_norot_sgrid.so bash tools/quick_bench.sh direct_r02 --config $c
  done
done 2>&1 | tee $O/ab.txt
bash tools/quick_bench.sh headline
bash tools/quick_bench.sh c64 --config c64l6_grid16_1024x512 --no-twin
tools/pmc_profile.sh c32l4_grid16_1024x512_r03a --config c32l4_grid16_1024x512 > $O/pmc.txt 2>&1; tail -40 $O/pmc.txt
