#!/bin/bash
mkdir -p gpurun_out
timeout 2000 python -m pytest tests -m gpu -q > gpurun_out/shcells_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/shcells_pytest.log
tail -6 gpurun_out/shcells_pytest.log
for v in 0 1; do echo "== FVSRN_CELL_TABLE=$v"; FVSRN_CELL_TABLE=$v timeout 900 python tools/bench_shaded.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'], 'unshaded %.2f fd %.2f adjoint %.2f ms' % (d['unshaded_ms'], d['finite_differences_ms'], d['adjoint_ms']))"; done
