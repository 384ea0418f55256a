#!/bin/bash
# r03 call 6: blend-ahead (fvsrn_network_prepare) in the frame pipeline; whole suite with the rotating SGRID variant off again
O=gpurun_out/r03c6; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
timeout 900 python tools/stripe_efficiency.py > $O/stripe_efficiency.jsonl 2>$O/stripe.err; cat $O/stripe_efficiency.jsonl
FVSRN_BENCH_BLEND_AHEAD=0 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 > $O/stripe_efficiency_no_ahead.jsonl 2>>$O/stripe.err; cat $O/stripe_efficiency_no_ahead.jsonl
FVSRN_BENCH_BLEND_AHEAD=0 FVSRN_WORKING_GRIDS=1 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 > $O/stripe_efficiency_r02_layout.jsonl 2>>$O/stripe.err; cat $O/stripe_efficiency_r02_layout.jsonl
bash tools/quick_bench.sh c32grid --config c32l4_grid16_1024x512
