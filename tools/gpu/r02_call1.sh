#!/bin/bash
# round 2, GPU call 1: issue-model microbenchmarks, sanity of the shipped build, RTZ-convert variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c1; mkdir -p $O
timeout 300 tools/microbench/r02_issue > $O/r02_issue.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_main.txt
for cfg in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do
  tools/quick_bench.sh main --config $cfg >> $O/bench.txt
  for f in fv-srn_amd/ablate/*.so; do
    FVSRN_LIBRARY=$PWD/$f tools/quick_bench.sh $(basename $f .so | sed s/libfvsrn_//) --config $cfg >> $O/bench.txt
  done
done
for f in fv-srn_amd/ablate/*.so; do
  echo "== $f" >> $O/pytest_variants.txt
  FVSRN_LIBRARY=$PWD/$f python -m pytest tests/test_gpu_parity.py tests/test_fuzz_parity.py -m gpu -q 2>&1 | tail -15 >> $O/pytest_variants.txt
done
cat $O/bench.txt
