#!/bin/bash
# r05 GPU call 4: new tests (batch / multi-frame launches, pyrenderer render_stripes), stripe efficiency with frames per submit, off-path kernel lines (V4 A/B)
O=gpurun_out/r05c; mkdir -p $O
export FVSRN_TEST_PROGRESS=$PWD/$O/progress.log
timeout 900 python -m pytest tests/test_gpu_stripes.py tests/test_pyrenderer.py -m gpu -q -k "batch or extract_color_of or cell_tables or rccl or render_stripes or two_rank or composes" 2>&1 | tail -40 > $O/gputest.txt; tail -5 $O/gputest.txt
export GPU_MAX_HW_QUEUES=8
export FVSRN_STRIPE_WORLDS=8
for B in 1 8; do
  FVSRN_STRIPE_BATCH=$B timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 c32l4_grid16_1024x512 >> $O/stripe_eff.jsonl 2>> $O/err.txt
  FVSRN_STRIPE_EMULATE_GATHER=24,512,100 FVSRN_STRIPE_BATCH=$B timeout 300 python tools/stripe_efficiency.py c32l4_fourier_1024x512 >> $O/stripe_eff_standin.jsonl 2>> $O/err.txt
done
FVSRN_STRIPE_WORLDS=2,4,8 FVSRN_STRIPE_BATCH=8 timeout 600 python tools/stripe_efficiency.py >> $O/stripe_eff_all_b8.jsonl 2>> $O/err.txt
cat $O/stripe_eff.jsonl $O/stripe_eff_standin.jsonl $O/stripe_eff_all_b8.jsonl
# off-path kernels: gather path by close-up / BYTE_GAUSSIAN, pipelined (default) vs fragment-major (FVSRN_OVERLAP_KERNEL=1) vs one wave per SIMD (variant library)
B="python bench.py --steps 12 --warmup 3 --no-twin --no-cpu-baseline"
for cfg in c32l4_grid16_1024x512 c64l6_grid16_1024x512; do
  for var in "--grid-encoding byte_linear" "--grid-encoding byte_gaussian" "--camera-distance 0.8" "--camera-distance 0.8 --grid-encoding byte_gaussian"; do
    for mode in default overlap w1; do
      env=""; [ $mode = overlap ] && env="FVSRN_OVERLAP_KERNEL=1"; [ $mode = w1 ] && env="FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_w1.so"
      [ $mode = w1 ] && [ $cfg = c32l4_grid16_1024x512 ] && continue
      echo "== $cfg $var $mode" >> $O/offpath.txt
      env $env $B --config $cfg $var 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','kernel','launch','variant')}))" >> $O/offpath.txt
    done
  done
done
for gm in finite_differences adjoint; do
  echo "== c64l6_grid16_1024x512 $gm" >> $O/offpath.txt
  $B --steps 4 --warmup 1 --config c64l6_grid16_1024x512 --gradient-mode $gm 2>> $O/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps({k:d[k] for k in ('value','ms_per_step','kernel','launch','variant')}))" >> $O/offpath.txt
done
cat $O/offpath.txt; tail -5 $O/err.txt
