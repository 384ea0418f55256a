#!/bin/bash
# r05 GPU call 10: the suite after the last kernel changes (fragment-major render_kernel at 48 / 64 channels, BYTE_GAUSSIAN decode order, batches with per-frame times on two lanes),
# the BYTE_GAUSSIAN lines, configs[4] stripes
O=gpurun_out/r05g; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gputest.txt; tail -4 $O/gputest.txt
export GPU_MAX_HW_QUEUES=8
B="python bench.py --no-twin --no-cpu-baseline"
pick='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernel","launch")}), d["roofline"]["frac"])'
for cfg in c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512; do
  echo "== $cfg byte_gaussian" >> $O/lines.txt; $B --config $cfg --grid-encoding byte_gaussian 2>> $O/err.txt | tee $O/bench_r05_${cfg}_byte_gaussian.json | python -c "$pick" >> $O/lines.txt
done
echo "== c64l6 gather path" >> $O/lines.txt; FVSRN_CELL_TABLE=0 $B --config c64l6_grid16_1024x512 2>> $O/err.txt | tee $O/bench_r05_c64l6_grid16_1024x512_gather_path.json | python -c "$pick" >> $O/lines.txt
FVSRN_STRIPE_WORLDS=2,4,8 FVSRN_STRIPE_BATCH=8 timeout 600 python tools/stripe_efficiency.py c64l6_grid16_time16_1024x512 > $O/stripe_eff_time16_b8.jsonl 2>> $O/err.txt
cat $O/lines.txt $O/stripe_eff_time16_b8.jsonl; tail -3 $O/err.txt
