#!/bin/bash
# builds tools/dev/bin/tap_repro: grid_tap of srn_device.hpp as it is compiled with FVSRN_TAP_NOPS = 0, and a textual copy of the same
# function with every stage spaced (grid_tap_spaced)
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d)
python3 - "$T" <<'PY'
import re, sys
s = open("fv-srn_amd/csrc/srn_device.hpp").read()
i = s.index("__device__ __forceinline__ GridTap grid_tap(const NetParams& P, float px, float py, float pz) {")
j = s.index("\n}\n", i) + 3
body = s[i:j].replace("GridTap grid_tap(", "GridTap grid_tap_spaced(").replace("FVSRN_TAP_NOPS", "127")
open(sys.argv[1] + "/grid_tap_spaced.hpp", "w").write("namespace fvsrn {\n" + body + "}\n")
PY
mkdir -p tools/dev/bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I fv-srn_amd/csrc -I $T tools/dev/tap_repro.hip -o tools/dev/bin/tap_repro
rm -rf $T
