#!/bin/bash
# usage: tools/kernel_resources.sh <object.o> [name-regex]: VGPR / SGPR / LDS / scratch of the gfx950 kernels in a hipcc object
set -eu
OBJ=$1; RE=${2:-.}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin $OBJ
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/dev.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s+(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    if re.search(r'''$RE''', name):
        print(f\"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>5s} spill {g('vgpr_spill_count')}\")
"
rm -rf $T
