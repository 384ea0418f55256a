#!/bin/bash
# r03 call 5: (1) rotating SGRID kernel, tuned, against the direct-feature kernel and both without their gathers (ablation);
# (2) evaluate_points at two batch sizes; (3) per-rank frame periods of configs[3] / [4] (StripeRenderer, two working grids vs one)
O=gpurun_out/r03c5; mkdir -p $O
A=$PWD/fv-srn_amd/ablate
for i in 1 2; do
  for v in rot2 direct rot2_noload direct_noload; do
    FVSRN_LIBRARY=$A/libfvsrn_$v.so bash tools/quick_bench.sh $v --config c32l4_grid16_1024x512
  done
done 2>&1 | tee $O/ab.txt
FVSRN_LIBRARY=$A/libfvsrn_rot2.so timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "resident or bench_networks or full_size" > $O/pytest_rot2.txt 2>&1; tail -3 $O/pytest_rot2.txt
timeout 900 python tools/bench_evaluate.py > $O/bench_evaluate.jsonl 2>$O/bench_evaluate.err; cut -c1-230 $O/bench_evaluate.jsonl
timeout 900 python tools/stripe_efficiency.py > $O/stripe_efficiency.jsonl 2>$O/stripe.err; cat $O/stripe_efficiency.jsonl
FVSRN_WORKING_GRIDS=1 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 > $O/stripe_efficiency_one_grid.jsonl 2>>$O/stripe.err; cat $O/stripe_efficiency_one_grid.jsonl
