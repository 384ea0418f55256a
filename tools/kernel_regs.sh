#!/bin/bash
# Developer tool (runs HERE): register / spill / LDS figures of the kernels in one object file of the build.
# usage: tools/kernel_regs.sh <object.o> [name filter (demangled substring)]
set -e
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fat.bin
$L/clang-offload-bundler --type=o --input=$T/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co --unbundle
$L/llvm-readelf --notes $T/dev.co | python3 -c "
import sys, re, subprocess
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = subprocess.run(['c++filt', g('name')], capture_output=True, text=True).stdout.strip()
    if flt in name:
        print('%-110s vgpr %s sgpr %s vgpr_spill %s sgpr_spill %s scratch %s' % (name[:110], g('vgpr_count'), g('sgpr_count'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('private_segment_fixed_size')))
" "$2"
rm -rf $T
