#!/bin/bash
# tools/disasm.sh <lib.so> <kernel name substring> : ISA of one kernel of a library build
LIB=$1; PAT=$2
T=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" $T/fat
# a library holds one fat binary per translation unit, back to back: split at the bundle magic
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
  /opt/rocm/lib/llvm/bin/llvm-objdump -d --no-show-raw-insn $f.co 2>/dev/null | awk -v pat="$PAT" '/^[0-9a-f]+ <.*>:/{p=index($0,pat)>0} p'
done
rm -rf $T
