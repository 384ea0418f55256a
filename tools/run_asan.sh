#!/bin/bash
# Host sanitizer run (CPU only): builds libfvsrn_asan.so (host objects with ASan + UBSan, fv-srn_amd/csrc/Makefile `asan`) and the
# sanitized oracle, then runs the CPU test-suite and the mutation fuzz of the .volnet / .cvol / scene-JSON parsers on them.
# usage: tools/run_asan.sh [mutations per seed]      -> profiles/r05/asan_report.txt
set -e
cd "$(dirname "$0")/.."
N=${1:-1000}
make -C fv-srn_amd/csrc -j8 asan 2>&1 | tail -1
make -C oracle asan 2>&1 | tail -1
ASAN=$(g++ -print-file-name=libasan.so)
# (libstdc++ next to libasan: ASan resolves __cxa_throw when it starts, before python has loaded any C++ library)
export LD_PRELOAD="$ASAN $(g++ -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export FVSRN_LIBRARY=$PWD/fv-srn_amd/libfvsrn_asan.so FVSRN_ORACLE_LIBRARY=$PWD/oracle/libsrn_oracle_asan.so
mkdir -p profiles/r05
{
  echo "# host sanitizer run: g++ -fsanitize=address,undefined on api.cpp launch_plan.cpp keyframes.cpp cvol_io.cpp scene_network.cpp pack.cpp + the C oracle; $(date -u +%F)"
  echo "## CPU test-suite (pytest -m 'not gpu', tests that load the C ABI / the oracle)"
  python -m pytest tests/test_volnet_format.py tests/test_capi_symbols.py tests/test_oracle_golden.py tests/test_oracle_models.py tests/test_grid_volume.py tests/test_torch_port.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -3
  echo "## mutation fuzz, $N mutations per seed input"
  python tools/host_fuzz.py $N 2>&1 | tail -5
} | tee profiles/r05/asan_report.txt
