#!/bin/bash
# new defaults (persistent stripe launches with 1/16 of the slots in reserve, eight hardware queues): a rank's frame period, with and
# without a stand-in for the collective; and the stripe tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_stripes.py tests/test_tiles_gloo.py -m gpu -x -q 2>&1 | tail -3
run() { echo "== $1"; shift; env "$@" python tools/stripe_efficiency.py $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   %s full %.2f ms; ' % (d['workload'], d['full_frame_ms']) + '; '.join('world %s: %.3f ms = %.1f %%' % (w, v['slowest_rank_frame_period_ms'], 100 * v['render_only_efficiency']) for w, v in d['world'].items()))"; }
C=""
run "defaults"                                    A=1
run "defaults + stand-in 24 x 512 x 150 us"       FVSRN_STRIPE_EMULATE_GATHER=24,512,150
run "reserve 0 + stand-in"                        FVSRN_STRIPE_EMULATE_GATHER=24,512,150 FVSRN_PERSISTENT_RESERVE=0
run "bounded waves (r02 default) + stand-in"      FVSRN_STRIPE_EMULATE_GATHER=24,512,150 FVSRN_PERSISTENT=0
run "four hardware queues + stand-in"             FVSRN_STRIPE_EMULATE_GATHER=24,512,150 GPU_MAX_HW_QUEUES=4
