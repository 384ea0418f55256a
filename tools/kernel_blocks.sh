#!/bin/bash
# Developer tool (runs HERE): per basic block of one kernel -- instructions, scratch loads / stores, MFMAs, v_dot2, global loads.
# usage: tools/kernel_blocks.sh <object.o> <mangled-name regex> [min block size]
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
$L/llvm-objdump -d --no-show-raw-insn $T/dev.co > $T/dis.s
n=$(grep -n "$2.*>:" $T/dis.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' $T/dis.s | awk '/^$/{exit} {print}' > $T/k.s
echo "listing: $T/k.s ($(wc -l < $T/k.s) lines)"
awk -v min=${3:-60} '{ if ($1 ~ /s_cbranch|s_branch|s_endpgm/) { if (tot >= min) printf "lines %5d-%5d %-16s instr %4d  scratch_load %3d scratch_store %3d  mfma %3d dot2 %3d vmem_load %3d lds %3d\n", first, NR, $1, tot, sl, ss, mf, d2, gl, ds; sl=ss=mf=d2=gl=tot=ds=0; first=NR+1 } else { tot++; if ($1 ~ /scratch_load/) sl++; if ($1 ~ /scratch_store/) ss++; if ($1 ~ /mfma/) mf++; if ($1 ~ /dot2/) d2++; if ($1 ~ /global_load|buffer_load/) gl++; if ($1 ~ /^ds_/) ds++; } }' $T/k.s
