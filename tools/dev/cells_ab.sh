#!/bin/bash
# Box script: the cell-table path (FVSRN_OPT_CELL_TABLE) against the gather path -- GPU suite, then the latent-grid bench lines both ways.
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/cells_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cells_pytest.log
tail -8 gpurun_out/cells_pytest.log
for cfg in c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do
for v in 0 1; do
  FVSRN_CELL_TABLE=$v timeout 600 python bench.py --config $cfg --no-cpu-baseline > gpurun_out/cells_bench_${cfg}_$v.json 2> gpurun_out/cells_bench_${cfg}_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cells_bench_${cfg}_$v.json").read().strip().splitlines()[-1])
print("$cfg cell_table=$v", d["value"]/1e9, "Gsamples/s", d["ms_per_step"], "ms", d["roofline"]["frac"], "twin", d["twin"]["value"]/1e9)
PY
done
done
