#!/bin/bash
# the driver's round-end sequence on the final tree: GPU suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "smoke" | cut -c1-220
python bench.py 2>/dev/null | cut -c1-700
