#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== default (bounded waves)"; python tools/dev/coschedule.py 2>/dev/null
echo "== persistent, no reserve"; FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=0 python tools/dev/coschedule.py 2>/dev/null
echo "== persistent, reserve 32"; FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=32 python tools/dev/coschedule.py 2>/dev/null
echo "== persistent, reserve 128"; FVSRN_PERSISTENT=1 FVSRN_PERSISTENT_RESERVE=128 python tools/dev/coschedule.py 2>/dev/null
