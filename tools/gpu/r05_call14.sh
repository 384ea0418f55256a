#!/bin/bash
# r05 GPU call 14: per-rank frame periods of the headline at world 8 for stripe heights 8 / 16 / 32 / 64 rows (is the slowest rank slow because of its rows?)
O=gpurun_out/r05k; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for rows in 16 8 32 64 16; do
  FVSRN_STRIPE_ROWS=$rows FVSRN_STRIPE_WORLDS=8 FVSRN_STRIPE_BATCH=8 timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 >> $O/stripe_rows.jsonl 2>> $O/err.txt
done
python - <<'PY'
import json
for l in open("gpurun_out/r05k/stripe_rows.jsonl"):
    d = json.loads(l); w = d["world"]["8"]
    print(d["workload"], "rows", w["stripe_rows"], "full %.3f" % d["full_frame_ms"], "eff %.3f" % w["render_only_efficiency"], w["rank_frame_period_ms"])
PY
tail -2 $O/err.txt
