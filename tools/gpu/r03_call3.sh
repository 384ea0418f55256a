#!/bin/bash
# r03 call 2: DEVICE-model fuzz (32 + outliers under pytest, 400-seed report), Gaussian TF variants, stripes, whole suite
O=gpurun_out/r03c3; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_stripes.py tests/test_fuzz_parity.py -q -m gpu > $O/new_tests.txt 2>&1; tail -12 $O/new_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_grid_volume.py tests/test_pyrenderer.py -q -m gpu -k "gaussian or Gaussian or evaluate_tf or preintegration" > $O/gauss.txt 2>&1; tail -12 $O/gauss.txt
timeout 1800 python tests/test_fuzz_parity.py 400 > $O/fuzz400.txt 2>$O/fuzz400.err; tail -2 $O/fuzz400.txt; grep -c FAIL $O/fuzz400.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
