#!/bin/bash
# r05 GPU call 15: is rank 0 the slowest rank of the headline at world 8 because it is measured first?
O=gpurun_out/r05l; mkdir -p $O
export GPU_MAX_HW_QUEUES=8
for rows in 16 8; do for order in forward reverse; do
  FVSRN_STRIPE_RANK_ORDER=$order FVSRN_STRIPE_ROWS=$rows FVSRN_STRIPE_WORLDS=8 FVSRN_STRIPE_BATCH=8 timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 2>> $O/err.txt | sed "s/^{/{\"order\": \"$order\", /" >> $O/stripe_order.jsonl
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r05l/stripe_order.jsonl"):
    d = json.loads(l); w = d["world"]["8"]
    print(d["order"], "rows", w["stripe_rows"], "full %.3f" % d["full_frame_ms"], "eff %.3f" % w["render_only_efficiency"], w["rank_frame_period_ms"])
PY
