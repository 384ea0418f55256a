#!/bin/bash
# adjoint + latent grid: piecewise-exact central differences -- parity, then the shaded / gradient benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c20; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "adjoint or gradient or shaded or curvature or batched" 2>&1 | tail -5 > $O/tests.txt
python tools/bench_shaded.py > $O/bench_shaded.jsonl 2>/dev/null
python tools/bench_evaluate_gradients.py > $O/bench_evaluate_gradients.jsonl 2>/dev/null
cat $O/tests.txt $O/bench_shaded.jsonl $O/bench_evaluate_gradients.jsonl
