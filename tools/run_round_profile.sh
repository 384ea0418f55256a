cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python bench.py > gpurun_out/bench_v12.json 2> gpurun_out/bench_v12.err
for c in c32l4_fourier_512x256 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_v12_$c.json 2>/dev/null; done
for c in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/pmc_profile.sh ${c}_v12 --config $c > /dev/null 2>&1; done
python tools/bench_evaluate.py > gpurun_out/bench_evaluate_v12.jsonl 2>/dev/null
for f in gpurun_out/bench_v12*.json; do python -c "
import json,sys
d=json.load(open('$f'))
print(d['config']['workload'].split(':')[0], '%.2f G %.3f ms frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac']), 'twin %.2f' % (d['twin']['value']/1e9) if d.get('twin') else '', d.get('cpu_baseline',{}).get('value'))"; done
