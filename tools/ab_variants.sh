#!/bin/bash
# Rebuilds libfvsrn.so with different build-time knobs ON THE GPU BOX and benches each (same device => comparable).
# usage: tools/ab_variants.sh "<EXTRA flags 1>" "<EXTRA flags 2>" ...
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for v in "$@"; do
  make -C fv-srn_amd/csrc clean >/dev/null 2>&1
  make -C fv-srn_amd/csrc -j16 EXTRA="$v" 2>&1 | grep -E "error" -A3
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('VARIANT [%s] relu %.2f Gs/s %.3f ms | snakealt %.2f Gs/s' % ('$v', d['value']/1e9, d['ms_per_step'], d['twin']['value']/1e9))"
done
make -C fv-srn_amd/csrc clean >/dev/null 2>&1
make -C fv-srn_amd/csrc -j16 2>&1 | grep -E "error" -A3
