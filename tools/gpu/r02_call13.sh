#!/bin/bash
# evaluate_small with grid + 32^3 grid bench row
mkdir -p gpurun_out/c13
python -m pytest tests -x -q -m gpu > gpurun_out/c13/pytest.txt 2>&1; tail -3 gpurun_out/c13/pytest.txt
python tools/bench_evaluate.py > gpurun_out/c13/evaluate.txt 2>&1; cat gpurun_out/c13/evaluate.txt | cut -c1-200
FVSRN_SMALL_KERNEL=0 python tools/bench_evaluate.py 2>&1 | grep grid16_relu | cut -c1-140
for c in c32l4_grid16_1024x512 c32l4_grid16r32_1024x512 c32l4_fourier_1024x512; do bash tools/quick_bench.sh base --config $c; done
