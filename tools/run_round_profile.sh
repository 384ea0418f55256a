cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-r02}
python -m pytest tests -m gpu -x -q 2>&1 | tail -1
python bench.py > gpurun_out/bench_${V}.json 2> gpurun_out/bench_${V}.err
for c in c32l4_fourier_512x256 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do python bench.py --config $c --no-cpu-baseline > gpurun_out/bench_${V}_$c.json 2>/dev/null; done
for c in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/pmc_profile.sh ${c}_${V} --config $c > /dev/null 2>&1; done
tools/pmc_profile.sh c32l4_fourier_snakealt_1024x512_${V} --config c32l4_fourier_1024x512 --activation SnakeAlt > /dev/null 2>&1
python - <<'PY'
import sys
sys.path.insert(0, '.')
import fvsrn_amd
from fvsrn_amd import synthetic, volnet_io
for act in ("ReLU", "SnakeAlt"):
    vn = synthetic.random_network(C=32, layers=4, activation=act, param=1.0, output_mode="density:direct", seed=1234, box_min=(-0.5, -0.5, -0.5))
    open("/tmp/protocol_%s.volnet" % act, "wb").write(volnet_io.save_volnet(vn))
PY
for a in ReLU SnakeAlt; do python tools/render_protocol.py /tmp/protocol_$a.volnet 2>/dev/null | tail -1 >> gpurun_out/protocol_c32l4_512x512_${V}.jsonl; done
python tools/bench_evaluate.py > /dev/null 2>&1; python tools/bench_evaluate.py > gpurun_out/bench_evaluate_${V}.jsonl 2>/dev/null
python tools/bench_grid_volume.py > gpurun_out/grid_volume_bench_${V}.json 2>/dev/null
python tools/bench_tail_variants.py 2>/dev/null | grep -v amdgpu > gpurun_out/tail_variants_${V}.txt
for f in gpurun_out/bench_${V}*.json; do python -c "
import json,sys
d=json.load(open('$f'))
print(d['config']['workload'].split(':')[0], '%.2f G %.3f ms frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac']), 'twin %.2f' % (d['twin']['value']/1e9) if d.get('twin') else '', d.get('cpu_baseline',{}).get('value'))"; done
