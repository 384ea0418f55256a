#!/bin/bash
O=gpurun_out/r03c10; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
for i in 1 2; do
FVSRN_BENCH_BLEND_AHEAD=1 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ahead', {k: round(v['slowest_rank_frame_period_ms'],3) for k,v in d['world'].items()})"
FVSRN_BENCH_BLEND_AHEAD=0 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('no ahead', {k: round(v['slowest_rank_frame_period_ms'],3) for k,v in d['world'].items()})"
FVSRN_BENCH_BLEND_AHEAD=0 FVSRN_WORKING_GRIDS=1 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('one grid', {k: round(v['slowest_rank_frame_period_ms'],3) for k,v in d['world'].items()})"
done | tee $O/ahead_ab.txt
