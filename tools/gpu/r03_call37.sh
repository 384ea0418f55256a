#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do echo "== run $rep"; python tools/dev/determinism.py 30 2>&1 | grep -v amdgpu.ids | grep "launches differ" | grep -v " 0 of"; done
for c in c32l4_grid16_1024x512 c64l6_grid16_1024x512 c32l4_fourier_1024x512; do python bench.py --config $c --no-cpu-baseline --no-twin 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config']['workload'].split(':')[0], '%.2f G  %.3f ms  frac %.3f' % (d['value'] / 1e9, d['ms_per_step'], d['roofline']['frac']))"; done
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
