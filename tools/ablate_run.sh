#!/bin/bash
# usage (GPU box): tools/ablate_run.sh [bench args]   -- benches libfvsrn.so and every fv-srn_amd/ablate/*.so
cd "$(dirname "$0")/.."
tools/quick_bench.sh full --no-twin "$@"
for f in fv-srn_amd/ablate/*.so; do
  FVSRN_LIBRARY=$PWD/$f tools/quick_bench.sh $(basename $f .so | sed s/libfvsrn_//) --no-twin "$@"
done
