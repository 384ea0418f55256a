#!/bin/bash
# adjoint evaluate tests + wave-priority variant of the LDS kernels
mkdir -p gpurun_out/c15
python -m pytest tests -x -q -m gpu > gpurun_out/c15/pytest.txt 2>&1; tail -3 gpurun_out/c15/pytest.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "adjoint_gradient_matches_oracle and evaluate" -s 2>&1 | grep "adjoint evaluate"
for c in c64l6_grid16_1024x512 c32l4_grid16_1024x512; do
  for i in 1 2; do
    bash tools/quick_bench.sh base --config $c
    FVSRN_SMALL_KERNEL=0 bash tools/quick_bench.sh base_nosmall --config $c --no-twin
    FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_ldsprio.so bash tools/quick_bench.sh ldsprio --config $c
    FVSRN_SMALL_KERNEL=0 FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_ldsprio.so bash tools/quick_bench.sh ldsprio_nosmall --config $c --no-twin
  done
done
FVSRN_SMALL_KERNEL=0 bash tools/quick_bench.sh base_nosmall --config c32l4_fourier_1024x512
FVSRN_SMALL_KERNEL=0 FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_ldsprio.so bash tools/quick_bench.sh ldsprio_nosmall --config c32l4_fourier_1024x512
