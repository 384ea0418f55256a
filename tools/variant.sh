#!/bin/bash
# Developer tool: builds fv-srn_amd/ablate/libfvsrn_<name>.so from the objects of the main build, recompiling only the listed
# translation units with extra flags (runs HERE: hipcc cross-compiles; tools/ablate_run.sh benches every variant on the GPU box).
# usage: tools/variant.sh <name> "<extra flags>" <tu> [<tu> ...]      e.g. tools/variant.sh rtz "-DFVSRN_CVT_RTZ=1" kernels_small_render
set -e
cd "$(dirname "$0")/../fv-srn_amd/csrc"
name=$1; flags=$2; shift 2
mkdir -p ../ablate build_var_$name
objs=""
for o in build/*.o; do
  b=$(basename $o .o); skip=0
  for tu in "$@"; do [ "$tu" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
pids=""
for tu in "$@"; do
  src=$tu.hip
  [ -f $src ] || src=$tu.cpp
  ./hipcc_fixed.sh $src build_var_$name/$tu.o -O3 -std=c++17 -fPIC $flags &   # (the same erratum pass as the main build)
  pids="$pids $!"
  objs="$objs build_var_$name/$tu.o"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../ablate/libfvsrn_$name.so $objs
rm -rf build_var_$name
ls -la ../ablate/libfvsrn_$name.so
