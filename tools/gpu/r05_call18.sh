#!/bin/bash
# r05 GPU call 18: NeRF-ladder networks in chain row order, double-angle features in evaluate_small_kernel: the suite, evaluate_points throughput, the headline line
O=gpurun_out/r05p; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gputest.txt; tail -4 $O/gputest.txt
for n in 1048576 4194304 16777216 67108864; do
  timeout 300 python tools/bench_evaluate.py $n 2>> $O/err.txt >> $O/bench_evaluate.jsonl
done
timeout 300 python tools/bench_evaluate.py 16777216 half 2>> $O/err.txt >> $O/bench_evaluate.jsonl
python - <<'PY'
import json
for l in open("gpurun_out/r05p/bench_evaluate.jsonl"):
    d = json.loads(l)
    print("%-52s n=2^%2d  %8.4f ms  %7.2f G points/s  mfma %.3f hbm %.3f" % (d["workload"], d["points"].bit_length() - 1, d["ms"], d["points_per_s"] / 1e9, d["roofline"]["mfma"]["frac"], d["roofline"]["hbm"]["frac"]))
PY
python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']
print('headline %.2f G frac %.4f single %.2f G twin %.2f exact %.2f' % (d['value']/1e9, r['frac'], d['single_frame_launches']['value']/1e9, d['twin']['value']/1e9, d['exact_features']['value']/1e9))"
for c in c32l4_grid16_1024x512 c64l6_grid16_1024x512; do python bench.py --config $c --no-cpu-baseline --no-twin 2>> $O/err.txt | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$c %.2f G frac %.4f' % (d['value']/1e9, d['roofline']['frac']))"; done
tail -3 $O/err.txt
