#!/bin/bash
# Developer tool (hazard bisect, r04): device assembly (.s, possibly patched by patch_s.py) -> code object -> fat binary -> replaces the
# .hip_fatbin section of an object hipcc built from the same translation unit (host side unchanged).
# usage: reasm.sh <device.s> <host-object-template.o> <out.o>
set -e
L=/opt/rocm/lib/llvm/bin
t=$(mktemp -d)
$L/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c "$1" -o $t/dev.o
$L/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o $t/dev.out $t/dev.o
$L/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$t/dev.out -output=$t/dev.hipfb
$L/llvm-objcopy --update-section .hip_fatbin=$t/dev.hipfb "$2" "$3"
rm -rf $t
