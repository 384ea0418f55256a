#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stripes.py tests/test_tiles_gloo.py -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" | cut -c1-300
run() { echo "== $1"; shift; env "$@" python tools/stripe_efficiency.py $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   %s full %.2f ms; ' % (d['workload'], d['full_frame_ms']) + '; '.join('world %s: %.3f ms = %.1f %%' % (w, v['slowest_rank_frame_period_ms'], 100 * v['render_only_efficiency']) for w, v in d['world'].items()))"; }
C="c64l6_grid16_1024x512"
run "defaults"                                    A=1
run "defaults + stand-in 24 x 512 x 150 us"       FVSRN_STRIPE_EMULATE_GATHER=24,512,150
run "defaults + stand-in 24 x 512 x 600 us"       FVSRN_STRIPE_EMULATE_GATHER=24,512,600
run "four queues + stand-in 150 us"               FVSRN_STRIPE_EMULATE_GATHER=24,512,150 GPU_MAX_HW_QUEUES=4
