#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/heur_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/heur_pytest.log
tail -5 gpurun_out/heur_pytest.log
python tools/dev/cell_footprint_sweep.py 2>&1 | grep -v amdgpu.ids | head -30
for c in c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/quick_bench.sh auto --config $c --no-twin; done
