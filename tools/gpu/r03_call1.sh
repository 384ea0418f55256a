#!/bin/bash
# r03 call 1: (a) may a GPU-initialised python process start a child interpreter?  (b) the new stripe / cross-stream tests,
# (c) the whole GPU suite, (d) smoke() with its 2-rank leg, (e) a first bench line
mkdir -p gpurun_out/r03c1
O=gpurun_out/r03c1
timeout 300 python - > $O/probe.txt 2>&1 <<'PY'
import subprocess, sys, torch
print("available", torch.cuda.is_available())
x = torch.ones(4, device="cuda").sum().item()
r = subprocess.run([sys.executable, "-c", "import torch; print('child', torch.cuda.is_available(), torch.ones(2, device='cuda').sum().item())"], capture_output=True, text=True, timeout=240)
print("child rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
PY
cat $O/probe.txt
timeout 1500 python -m pytest tests/test_gpu_stripes.py -x -q -m gpu > $O/stripes.txt 2>&1; tail -15 $O/stripes.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 900 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -4 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
timeout 600 python bench.py --config c64l6_grid16_time16_1024x512 --no-cpu-baseline --no-twin > $O/bench_time16.json 2>$O/bench_time16.err; cat $O/bench_time16.json
