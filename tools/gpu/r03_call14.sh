#!/bin/bash
O=gpurun_out/r03c14; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/stripe_efficiency.py > $O/stripe_efficiency.jsonl 2>$O/stripe.err; python - <<'PY'
import json
for l in open("gpurun_out/r03c14/stripe_efficiency.jsonl"):
    r=json.loads(l); print(r['workload'], round(r['full_frame_ms'],2), {k:(round(v['slowest_rank_frame_period_ms'],2), round(v['render_only_efficiency'],3)) for k,v in r['world'].items()})
PY
FVSRN_WORKING_GRIDS=1 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('one grid', round(r['full_frame_ms'],2), {k:(round(v['slowest_rank_frame_period_ms'],2), round(v['render_only_efficiency'],3)) for k,v in r['world'].items()})"
FVSRN_BENCH_BLEND_AHEAD=1 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('blend ahead', round(r['full_frame_ms'],2), {k:(round(v['slowest_rank_frame_period_ms'],2), round(v['render_only_efficiency'],3)) for k,v in r['world'].items()})"
timeout 600 python tools/dev/stress_concurrent.py 60 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-330 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | cut -c1-160
