#!/bin/bash
O=gpurun_out/r03c8; mkdir -p $O
timeout 600 python tools/dev/stress_concurrent.py 100 2>&1 | grep -v amdgpu.ids | cut -c1-60,150-330 | tee $O/stress_fenced.txt
timeout 600 python tools/dev/stress_ahead.py 40 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tee $O/stress_ahead_fenced.txt
timeout 900 python tools/stripe_efficiency.py > $O/stripe_efficiency.jsonl 2>$O/stripe.err; cat $O/stripe_efficiency.jsonl
