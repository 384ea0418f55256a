#!/bin/bash
# a rank's frame pipeline at world 8 with a stand-in for the all-gather (24 workgroups x 512 threads x 150 us on the comm stream)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FVSRN_STRIPE_WORLDS=${WORLDS:-8}
C=${CFG:-c64l6_grid16_1024x512}
run() { echo "== $1"; shift; env "$@" python tools/stripe_efficiency.py $C 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   full %.2f ms; ' % d['full_frame_ms'] + '; '.join('world %s: %.3f ms = %.1f %%' % (w, v['slowest_rank_frame_period_ms'], 100 * v['render_only_efficiency']) for w, v in d['world'].items()))"; }
G=24,512,150
run "no gather stand-in, default (one unit pair per wave)"   A=1
run "no gather stand-in, persistent"                         FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "stand-in, default"                                      FVSRN_STRIPE_EMULATE_GATHER=$G
run "stand-in, persistent, no reserve"                       FVSRN_STRIPE_EMULATE_GATHER=$G FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "stand-in, persistent, reserve 16"                       FVSRN_STRIPE_EMULATE_GATHER=$G FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=16
run "stand-in, persistent, reserve 32"                       FVSRN_STRIPE_EMULATE_GATHER=$G FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=32
run "stand-in, persistent, reserve 64"                       FVSRN_STRIPE_EMULATE_GATHER=$G FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=64
run "stand-in 400 us, default"                               FVSRN_STRIPE_EMULATE_GATHER=24,512,400
run "stand-in 400 us, persistent, no reserve"                FVSRN_STRIPE_EMULATE_GATHER=24,512,400 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0
run "stand-in 400 us, persistent, reserve 32"                FVSRN_STRIPE_EMULATE_GATHER=24,512,400 FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=32
