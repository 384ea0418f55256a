#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c4; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest.txt
tools/quick_bench.sh main --config c64l6_grid16_time16_1024x512 >> $O/bench.txt
FVSRN_KEYFRAME_SLOTS=2 tools/quick_bench.sh slots2 --config c64l6_grid16_time16_1024x512 >> $O/bench.txt
FVSRN_KEYFRAME_SLOTS=3 tools/quick_bench.sh slots3 --config c64l6_grid16_time16_1024x512 >> $O/bench.txt
cat $O/pytest.txt $O/bench.txt
