#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FVSRN_STRIPE_WORLDS=8
C=c64l6_grid16_1024x512
run() { echo "== $1"; shift; env "$@" python tools/stripe_efficiency.py $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   full %.2f ms; ' % d['full_frame_ms'] + '; '.join('world %s: %.3f ms = %.1f %%' % (w, v['slowest_rank_frame_period_ms'], 100 * v['render_only_efficiency']) for w, v in d['world'].items()))"; }
run "stand-in 1 us, default"                        FVSRN_STRIPE_EMULATE_GATHER=24,512,1
run "stand-in 1 us, persistent"                     FVSRN_STRIPE_EMULATE_GATHER=24,512,1 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "stand-in 150 us, persistent"                   FVSRN_STRIPE_EMULATE_GATHER=24,512,150 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "8 hw queues: stand-in 150 us, default"         GPU_MAX_HW_QUEUES=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,150
run "8 hw queues: stand-in 150 us, persistent"      GPU_MAX_HW_QUEUES=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,150 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "8 hw queues: stand-in 400 us, persistent"      GPU_MAX_HW_QUEUES=8 FVSRN_STRIPE_EMULATE_GATHER=24,512,400 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "8 hw queues: no stand-in, default"             GPU_MAX_HW_QUEUES=8
run "8 hw queues: no stand-in, persistent"          GPU_MAX_HW_QUEUES=8 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
