#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c8; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest.txt
for r in 1 2; do for cfg in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/quick_bench.sh main --config $cfg >> $O/bench.txt; done; done
python tools/bench_evaluate.py > /dev/null 2>&1; python tools/bench_evaluate.py > $O/bench_evaluate.jsonl 2>/dev/null
tail -8 $O/pytest.txt; cat $O/bench.txt; cut -c1-300 $O/bench_evaluate.jsonl
