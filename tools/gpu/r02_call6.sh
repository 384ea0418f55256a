#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c6; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/pytest.txt
for cfg in c32l4_fourier_1024x512 c32l4_grid16_1024x512 c64l6_grid16_1024x512; do tools/quick_bench.sh main --config $cfg >> $O/bench.txt; done
cat $O/pytest.txt $O/bench.txt
