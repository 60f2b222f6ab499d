#!/bin/bash
# VGPRs, SGPRs, scratch bytes and LDS bytes of every kernel of a library build (code object notes): tools/kernel_regs.sh [lib.so]
LIB=${1:-realtimepathtracingresearchframework_amd/librptr_hip.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" /dev/stdout) --output=$T/co --unbundle 2>/dev/null \
  || { /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat; /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat --output=$T/co --unbundle; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/co | python3 -c "
import sys,re
txt=sys.stdin.read()
for m in re.finditer(r'\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)', txt, re.S):
    pass
# entries are yaml maps; parse loosely
blocks=txt.split('- .agpr_count')
for b in blocks[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', b) or [None,'?'])[1]
    print('%-70s vgpr %3s sgpr %3s scratch %5s lds %6s' % (g('name')[:70], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
rm -rf $T
