#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FVSRN_NO_SCRATCH_FENCE=1
run() { echo "== $1"; shift; env "$@" python tools/stripe_efficiency.py c64l6_grid16_1024x512 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l)
    print('   %s full %.2f ms; ' % (d['workload'], d['full_frame_ms']) + '; '.join('world %s: %.3f ms = %.1f %%' % (w, v['slowest_rank_frame_period_ms'], 100 * v['render_only_efficiency']) for w, v in d['world'].items()))"; }
run "no fence, stripes with render_stripe_kernel (default)"   A=1
run "no fence, stripes with render_kernel"                    FVSRN_OVERLAP_KERNEL=0
run "no fence, render_kernel + stand-in"                      FVSRN_OVERLAP_KERNEL=0 FVSRN_STRIPE_EMULATE_GATHER=24,512,150
run "no fence, render_stripe_kernel + stand-in"               FVSRN_STRIPE_EMULATE_GATHER=24,512,150
