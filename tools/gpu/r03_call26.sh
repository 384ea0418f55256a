#!/bin/bash
# what limits a rank's share at world 8 (configs[3])?  one GPU, no collective
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export FVSRN_STRIPE_WORLDS=8
C=c64l6_grid16_1024x512
echo "default";            python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "persistent";         FVSRN_PERSISTENT=1 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "segments 1";         FVSRN_SEGMENTS=1 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "segments 2";         FVSRN_SEGMENTS=2 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "segments 4";         FVSRN_SEGMENTS=4 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "segments 8";         FVSRN_SEGMENTS=8 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "no pipeline";        FVSRN_BENCH_PIPELINE=0 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
echo "overlap kernel off"; FVSRN_OVERLAP_KERNEL=0 python tools/stripe_efficiency.py $C 2>/dev/null | cut -c1-400
