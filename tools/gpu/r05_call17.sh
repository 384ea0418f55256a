#!/bin/bash
# r05 GPU call 17: 8-row stripes as bench.py's default: stripe tests, smoke (two gloo ranks on one GPU), the RCCL route with a one-rank group, the tool's full record
O=gpurun_out/r05n; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 900 python -m pytest tests/test_gpu_stripes.py tests/test_pyrenderer.py -m gpu -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" > $O/smoke_r05.txt; tail -2 $O/smoke_r05.txt | cut -c1-300
export GPU_MAX_HW_QUEUES=8
python bench.py --force-collective --no-twin --no-cpu-baseline > $O/bench_r05_force_collective_nccl.json 2>> $O/err.txt
python bench.py --force-collective --no-twin --no-cpu-baseline --gather root --payload rgba8 > $O/bench_r05_force_collective_nccl_root_rgba8.json 2>> $O/err.txt
python bench.py --force-collective --no-twin --no-cpu-baseline --config c64l6_grid16_time16_1024x512 > $O/bench_r05_force_collective_nccl_c64l6_time16.json 2>> $O/err.txt
for f in $O/bench_r05_force*.json; do python -c "
import json; d=json.load(open('$f')); print('$f'.split('/')[-1], '%.2f G' % (d['value']/1e9), d.get('frame_check'), d['config'].get('parallelism'), d.get('per_rank'))"; done
FVSRN_STRIPE_BATCH=8 timeout 900 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512 > $O/stripe_efficiency_r05.jsonl 2>> $O/err.txt
FVSRN_STRIPE_BATCH=1 timeout 900 python tools/stripe_efficiency.py > $O/stripe_efficiency_frame_by_frame_r05.jsonl 2>> $O/err.txt
FVSRN_STRIPE_ROWS=16 FVSRN_STRIPE_BATCH=8 timeout 600 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512 > $O/stripe_efficiency_16_rows_r05.jsonl 2>> $O/err.txt
if [ -f tools/dev/bin/liboccupy.so ]; then
  FVSRN_STRIPE_BATCH=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,100 timeout 600 python tools/stripe_efficiency.py > $O/stripe_efficiency_with_stand_in_r05.jsonl 2>> $O/err.txt
fi
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05n/stripe_efficiency*.jsonl")):
    for l in open(f):
        d = json.loads(l)
        print(f.split("/")[-1][18:-6], d["workload"], "full %.3f" % d["full_frame_ms"], " ".join("w%s %.3f (host %.1f us)" % (w, v["render_only_efficiency"], v["host_us_per_frame"]) for w, v in d["world"].items()))
PY
tail -2 $O/err.txt
