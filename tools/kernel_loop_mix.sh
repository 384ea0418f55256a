#!/bin/bash
# Developer tool (runs HERE): static instruction mix of one kernel of an object file.
# usage: tools/kernel_loop_mix.sh <object.o> <mangled-name regex> [first line] [last line]   -> prints the listing path and the mix
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
$L/llvm-objdump -d --no-show-raw-insn $T/dev.co > $T/dis.s
n=$(grep -n "$2.*>:" $T/dis.s | head -1 | cut -d: -f1)
awk -v n=$n 'NR>=n' $T/dis.s | awk '/^$/{exit} {print}' > $T/k.s
a=${3:-1}; b=${4:-$(wc -l < $T/k.s)}
echo "listing: $T/k.s ($(wc -l < $T/k.s) lines), mix of lines $a..$b"
sed -n "${a},${b}p" $T/k.s | awk '{print $1}' | grep -v "^$" | sort | uniq -c | sort -rn | head -40
