#!/bin/bash
# r05 GPU call 11: evaluate_points in one launch (in-kernel unclamped pass for batches with a point outside the box) + fp16 I/O: tests, then throughput
O=gpurun_out/r05h; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_pyrenderer.py tests/test_reference_matrix.py -m gpu -q -x -k "evaluate or half or matrix or golden" 2>&1 | tail -15 > $O/gputest_eval.txt; tail -6 $O/gputest_eval.txt
for n in 1048576 4194304 16777216 67108864; do
  timeout 300 python tools/bench_evaluate.py $n 2>> $O/err.txt >> $O/bench_evaluate.jsonl
  timeout 300 python tools/bench_evaluate.py $n half 2>> $O/err.txt >> $O/bench_evaluate.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/r05h/bench_evaluate.jsonl"):
    d = json.loads(l)
    print("%-52s n=2^%2d  %8.4f ms  %7.2f G points/s  mfma %.3f hbm %.3f" % (d["workload"], d["points"].bit_length() - 1, d["ms"], d["points_per_s"] / 1e9, d["roofline"]["mfma"]["frac"], d["roofline"]["hbm"]["frac"]))
PY
tail -3 $O/err.txt
