#!/bin/bash
O=gpurun_out/r03c13; mkdir -p $O
A=$PWD/fv-srn_amd/ablate/libfvsrn_kmajor_grid.so
for i in 1 2; do
  bash tools/quick_bench.sh pipelined --config c64l6_grid16_1024x512
  FVSRN_LIBRARY=$A bash tools/quick_bench.sh kmajor --config c64l6_grid16_1024x512
done 2>&1 | tee $O/ab.txt
for i in 1 2; do
python tools/stripe_efficiency.py c64l6_grid16_1024x512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('pipelined+fence', round(d['full_frame_ms'],2), {k: round(v['slowest_rank_frame_period_ms'],3) for k,v in d['world'].items()})"
FVSRN_LIBRARY=$A python tools/stripe_efficiency.py c64l6_grid16_1024x512 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('kmajor no scratch', round(d['full_frame_ms'],2), {k: round(v['slowest_rank_frame_period_ms'],3) for k,v in d['world'].items()})"
done | tee -a $O/ab.txt
