#!/bin/bash
# VGPRs, SGPRs, scratch bytes and LDS bytes of every kernel of a library build (code object notes; one fat binary per translation
# unit): tools/kernel_regs.sh [lib.so]
LIB=${1:-realtimepathtracingresearchframework_amd/librptr_hip.so}
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat
python3 - "$T" <<'PY'
import sys
d=sys.argv[1]; b=open(d+'/fat','rb').read(); magic=b'__CLANG_OFFLOAD_BUNDLE__'
pos=[]; i=b.find(magic)
while i>=0: pos.append(i); i=b.find(magic,i+1)
for k,p in enumerate(pos):
    open('%s/fat%d'%(d,k),'wb').write(b[p:(pos[k+1] if k+1<len(pos) else len(b))])
PY
for f in $T/fat[0-9]*; do
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$f --output=$f.co --unbundle 2>/dev/null || continue
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for b in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s+(\S+)', b) or [None,'?'])[1]
    print('%-70s vgpr %3s sgpr %3s scratch %5s lds %6s' % (g('name')[:70], g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
"
done
rm -rf $T
