#!/bin/bash
# r05 GPU call 5: tests touched by the last changes, stripe efficiency with batches and no reserve, BYTE_GAUSSIAN after the frac == 0 path, frames per launch at world 1
O=gpurun_out/r05d; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 1200 python -m pytest tests/test_gpu_stripes.py tests/test_pyrenderer.py tests/test_gpu_parity.py -m gpu -q -k "not row_band" 2>&1 | tail -30 > $O/gputest.txt; tail -5 $O/gputest.txt
export GPU_MAX_HW_QUEUES=8
FVSRN_STRIPE_WORLDS=2,4,8 FVSRN_STRIPE_BATCH=8 timeout 900 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512 c64l6_grid16_time16_1024x512 > $O/stripe_eff_b8.jsonl 2>> $O/err.txt
FVSRN_STRIPE_WORLDS=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,100 FVSRN_STRIPE_BATCH=8 timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 > $O/stripe_eff_b8_standin.jsonl 2>> $O/err.txt
FVSRN_STRIPE_WORLDS=8 FVSRN_STRIPE_BATCH=1 timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 > $O/stripe_eff_b1.jsonl 2>> $O/err.txt
cat $O/stripe_eff_b8.jsonl $O/stripe_eff_b8_standin.jsonl $O/stripe_eff_b1.jsonl
B="python bench.py --steps 12 --warmup 3 --no-twin --no-cpu-baseline"
pick='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernel","launch","host_us_per_frame")}), d["roofline"]["frac"])'
for cfg in c32l4_grid16_1024x512 c64l6_grid16_1024x512; do
  echo "== $cfg byte_gaussian" >> $O/lines.txt; $B --config $cfg --grid-encoding byte_gaussian 2>> $O/err.txt | python -c "$pick" >> $O/lines.txt
done
echo "== c64l6_grid16_time16 byte_gaussian (frac != 0)" >> $O/lines.txt; $B --config c64l6_grid16_time16_1024x512 --grid-encoding byte_gaussian 2>> $O/err.txt | python -c "$pick" >> $O/lines.txt
for K in 1 2 4 8; do
  for cfg in c32l4_fourier_1024x512 c32l4_fourier_512x256; do
    echo "== $cfg frames-per-submit $K" >> $O/lines.txt; $B --steps 32 --warmup 8 --config $cfg --frames-per-submit $K 2>> $O/err.txt | python -c "$pick" >> $O/lines.txt
  done
done
echo "== headline pipelined (two streams), K=1" >> $O/lines.txt; FVSRN_BENCH_PIPELINE=1 $B --steps 32 --warmup 8 2>> $O/err.txt | python -c "$pick" >> $O/lines.txt
cat $O/lines.txt; tail -5 $O/err.txt
