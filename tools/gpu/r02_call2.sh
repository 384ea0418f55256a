#!/bin/bash
# round 2, GPU call 2: new parity tests on the r01 binary; no-packed-fp32 variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c2; mkdir -p $O
python -m pytest tests -m gpu -q -x --deselect tests/test_fuzz_parity.py 2>&1 | tail -15 > $O/pytest_main.txt
python tests/test_fuzz_parity.py 32 > $O/fuzz_report.txt 2>&1
for cfg in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do
  tools/quick_bench.sh main --config $cfg >> $O/bench.txt
  for f in fv-srn_amd/ablate/*.so; do
    FVSRN_LIBRARY=$PWD/$f tools/quick_bench.sh $(basename $f .so | sed s/libfvsrn_//) --config $cfg >> $O/bench.txt
  done
done
# second pass in reverse order (clock / box drift)
for cfg in c32l4_fourier_1024x512; do
  for f in $(ls -r fv-srn_amd/ablate/*.so); do
    FVSRN_LIBRARY=$PWD/$f tools/quick_bench.sh $(basename $f .so | sed s/libfvsrn_//) --config $cfg >> $O/bench.txt
  done
  tools/quick_bench.sh main --config $cfg >> $O/bench.txt
done
FVSRN_LIBRARY=$PWD/fv-srn_amd/ablate/libfvsrn_nopk.so python -m pytest tests -m gpu -q --deselect tests/test_fuzz_parity.py 2>&1 | tail -8 > $O/pytest_nopk.txt
cat $O/bench.txt $O/pytest_main.txt $O/pytest_nopk.txt
