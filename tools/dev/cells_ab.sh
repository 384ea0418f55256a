#!/bin/bash
# Box script: the cell-table path (FVSRN_OPT_CELL_TABLE) against the gather path -- GPU suite, then the 32x4 + 16^3 grid bench line both ways.
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/cells_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/cells_pytest.log
tail -8 gpurun_out/cells_pytest.log
for v in 0 1; do
  FVSRN_CELL_TABLE=$v timeout 600 python bench.py --config c32l4_grid16_1024x512 --no-cpu-baseline > gpurun_out/cells_bench_$v.json 2> gpurun_out/cells_bench_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/cells_bench_$v.json").read().strip().splitlines()[-1])
print("cell_table=$v", d["value"]/1e9, "Gsamples/s", d["ms_per_step"], "ms", d["roofline"]["frac"], "twin", d["twin"]["value"]/1e9, "exact", d["exact_features"]["value"]/1e9)
PY
done
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/cells_bench_headline.json 2> gpurun_out/cells_bench_headline.err
python - <<PY
import json
d=json.loads(open("gpurun_out/cells_bench_headline.json").read().strip().splitlines()[-1])
print("headline", d["value"]/1e9, "Gsamples/s", d["ms_per_step"], "ms", d["roofline"]["frac"], "twin", d["twin"]["value"]/1e9, "exact", d["exact_features"]["value"]/1e9)
PY
