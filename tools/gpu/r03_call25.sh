#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "evaluate or golden or g1 or g2 or fixture" 2>&1 | tail -3
python tools/bench_evaluate.py > /dev/null 2>&1; python tools/bench_evaluate.py 2>/dev/null | cut -c1-170
