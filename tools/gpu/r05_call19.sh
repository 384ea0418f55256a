#!/bin/bash
# r05 GPU call 19: the rebuilt (reverted) binary = the committed sources: suite, smoke, the driver's bench command, evaluate_points
O=gpurun_out/r05q; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gputest.txt; tail -3 $O/gputest.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke ok" | cut -c1-200
python bench.py > $O/bench_r05.json 2>> $O/err.txt; python -c "
import json; d=json.load(open('$O/bench_r05.json')); r=d['roofline']
print('headline %.2f G frac %.4f (%.4f/%.4f/%.4f) single %.2f G twin %.2f exact %.2f cpu %.3g' % (d['value']/1e9, r['frac'], r['mfma_frac_min'], r['mfma_frac_median'], r['mfma_frac_max'], d['single_frame_launches']['value']/1e9, d['twin']['value']/1e9, d['exact_features']['value']/1e9, d['cpu_baseline']['value']))"
python tools/bench_evaluate.py 16777216 > $O/bench_evaluate.jsonl 2>> $O/err.txt; python -c "
import json
for l in open('$O/bench_evaluate.jsonl'):
    d=json.loads(l); print(d['workload'], '%.1f G points/s' % (d['points_per_s']/1e9))"
