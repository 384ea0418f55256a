#!/bin/bash
# round 2, GPU call 3: full GPU suite on the rebuilt library (options API), fuzz report with per-step features, bench of all configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c3; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
python tests/test_fuzz_parity.py 32 2>&1 | cut -c1-200 > $O/fuzz_report.txt
for cfg in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/quick_bench.sh main --config $cfg >> $O/bench.txt; done
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
FVSRN_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 8 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err
cat $O/pytest.txt $O/bench.txt $O/smoke.txt; tail -3 $O/bench_gloo2.err; cut -c1-600 $O/bench_gloo2.json
