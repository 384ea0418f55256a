#!/bin/bash
O=gpurun_out/r03c11; mkdir -p $O
tools/microbench/bin/r03_snakealt 2>&1 | grep -v amdgpu.ids | tee $O/snakealt_microbench.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "evaluate" > $O/pytest_eval.txt 2>&1; tail -2 $O/pytest_eval.txt
timeout 900 python tools/bench_evaluate.py > $O/bench_evaluate.jsonl 2>$O/bench_evaluate.err; python - <<'PY'
import json
for l in open("gpurun_out/r03c11/bench_evaluate.jsonl"):
    d = json.loads(l); print(d["workload"], d["points"], "%.1f G points/s" % (d["points_per_s"]/1e9), d["kernel"], "mfma %.3f hbm %.3f" % (d["roofline"]["mfma"]["frac"], d["roofline"]["hbm"]["frac"]))
PY
python tools/bench_shaded.py 2>/dev/null | tail -8
