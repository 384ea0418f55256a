#!/bin/bash
# r03 call 7: evaluate_points with the re-scaled images (two-launch ReLU scheme), lean index arithmetic
O=gpurun_out/r03c7; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_pyrenderer.py tests/test_gpu_stripes.py -q -m gpu -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/bench_evaluate.py 16777216 > $O/bench_evaluate.jsonl 2>$O/bench_evaluate.err; python - <<'PY'
import json
for l in open("gpurun_out/r03c7/bench_evaluate.jsonl"):
    d = json.loads(l); print(d["workload"], d["points"], "%.1f G points/s" % (d["points_per_s"]/1e9), d["kernel"], "mfma %.3f hbm %.3f" % (d["roofline"]["mfma"]["frac"], d["roofline"]["hbm"]["frac"]))
PY
FVSRN_DISABLE_RELU_CLAMP=1 timeout 600 python tools/bench_evaluate.py 16777216 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d = json.loads(l); print('plain image:', d['workload'], '%.1f G points/s' % (d['points_per_s']/1e9), d['kernel'])"
