#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
tools/dev/bin/grid_diff_unit 2>&1 | grep "^step"
