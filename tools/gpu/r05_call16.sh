#!/bin/bash
# r05 GPU call 16: tools/stripe_efficiency.py with a spin-up per pipeline: all workloads, eight poses per launch; stripe heights 8 and 16; rank order both ways for the headline
O=gpurun_out/r05m; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for order in forward reverse; do
  FVSRN_STRIPE_RANK_ORDER=$order FVSRN_STRIPE_WORLDS=8 FVSRN_STRIPE_BATCH=8 timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 2>> $O/err.txt | sed "s/^{/{\"order\": \"$order\", /" >> $O/stripe_order_spinup.jsonl
done
FVSRN_STRIPE_BATCH=8 timeout 900 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512 > $O/stripe_efficiency_r05.jsonl 2>> $O/err.txt
FVSRN_STRIPE_ROWS=8 FVSRN_STRIPE_BATCH=8 timeout 600 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 > $O/stripe_efficiency_8_rows_r05.jsonl 2>> $O/err.txt
python - <<'PY'
import json
for f in ("stripe_order_spinup", "stripe_efficiency_r05", "stripe_efficiency_8_rows_r05"):
    for l in open("gpurun_out/r05m/%s.jsonl" % f):
        d = json.loads(l)
        for w, v in d["world"].items():
            print(f[:22], d.get("order", ""), d["workload"], "world", w, "rows", v["stripe_rows"], "full %.3f" % d["full_frame_ms"], "eff %.3f" % v["render_only_efficiency"], v["rank_frame_period_ms"])
PY
