#!/bin/bash
# r05 GPU call 13: the suite on the one-launch evaluate_points build (in-kernel unclamped pass, fp16 I/O, dwordx3 / two-buffer fetch, phases ahead), then its throughput
O=gpurun_out/r05j; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gputest.txt; tail -4 $O/gputest.txt
for n in 1048576 4194304 16777216 67108864; do
  timeout 300 python tools/bench_evaluate.py $n 2>> $O/err.txt >> $O/bench_evaluate.jsonl
done
timeout 300 python tools/bench_evaluate.py 16777216 half 2>> $O/err.txt >> $O/bench_evaluate.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05j/bench_evaluate.jsonl"):
    d = json.loads(l)
    print("%-52s n=2^%2d  %8.4f ms  %7.2f G points/s  mfma %.3f hbm %.3f" % (d["workload"], d["points"].bit_length() - 1, d["ms"], d["points_per_s"] / 1e9, d["roofline"]["mfma"]["frac"], d["roofline"]["hbm"]["frac"]))
PY
tail -3 $O/err.txt
