#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep -v " 0 of"
python tools/dev/stress_concurrent.py 100 2>&1 | grep -v amdgpu.ids | grep -c "differing frames: {}"
C="c64l6_grid16_1024x512" ; python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-420
