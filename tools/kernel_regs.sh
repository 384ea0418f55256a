#!/bin/bash
# Developer tool (runs HERE): VGPRs / spills / scratch of every kernel in one object.   usage: tools/kernel_regs.sh <object.o> [name regex]
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d); trap 'rm -rf $T' EXIT
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
$L/llvm-readelf --notes $T/dev.co | grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_count|sgpr_spill_count|private_segment_fixed_size|agpr_count):" | \
  awk '/\.name:/{if(n)print n, a; n=$2; a=""} !/\.name:/{a=a" "$1$2} END{print n, a}' | grep -E "${2:-.}" | sed 's/_ZN5fvsrn//; s/EvNS_9NetParams.*Py//'
