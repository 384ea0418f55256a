#!/bin/bash
# Compiles ONE .hip / .cpp translation unit for gfx950 with the packed-fp32 erratum pass between the compiler and the assembler:
#   device code -> assembly (hipcc -S) -> tools/fix_pk_opsel.py (no v_pk_*_f32 with op_sel:[0,1], profiles/r04/nondeterminism_r04.md)
#   -> code object -> fat binary, then the host side with that binary embedded.  Same object layout as `hipcc -c`.
# usage: hipcc_fixed.sh <source> <object> [compiler flags ...]          (FVSRN_KEEP_ASM=1 keeps <object>.s next to the object)
set -e
src=$1; obj=$2; shift 2
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
# The kernels are written for gfx950 only (MFMA shapes, LDS size, the erratum this pass exists for): any other ARCH from the Makefile is an error, not a
# silent gfx950 build.
ARCH=${ARCH:-gfx950}
[ "$ARCH" = gfx950 ] || { echo "hipcc_fixed.sh: ARCH=$ARCH, but this code base targets gfx950 (MI355X) only" >&2; exit 1; }
L=/opt/rocm/lib/llvm/bin
here=$(cd "$(dirname "$0")" && pwd)
x=""; case "$src" in *.cpp) x="-x hip";; esac
t=$(mktemp -d)
trap 'rm -rf "$t"' EXIT
$HIPCC --offload-arch=gfx950 "$@" $x --cuda-device-only -S "$src" -o $t/dev.s 2> >(grep -v "argument unused during compilation: '--hip-link'" >&2)
python3 "$here/../../tools/fix_pk_opsel.py" $t/dev.s $t/dev.fixed.s > $t/fix.log || { cat $t/fix.log; exit 1; }
[ -n "$FVSRN_KEEP_ASM" ] && cp $t/dev.fixed.s "$obj.s"
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $t/dev.fixed.s -o $t/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $t/dev.out $t/dev.o
# what was assembled, looked at once more: an instruction with the bad selection in a spelling the rewrite does not match must not reach the binary
$L/llvm-objdump -d --no-show-raw-insn $t/dev.out > $t/dev.dis
python3 "$here/../../tools/fix_pk_opsel.py" --check $t/dev.dis > $t/check.log || { cat $t/check.log; echo "hipcc_fixed.sh: $src still holds a packed-fp32 instruction with op_sel:[0,1]" >&2; exit 1; }
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$t/dev.out -output=$t/dev.hipfb
# (-MMD: the headers this translation unit includes, for the Makefile)
$HIPCC --offload-arch=gfx950 "$@" $x --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $t/dev.hipfb -MMD -MF "$obj.d" -MT "$obj" -c "$src" -o "$obj"
