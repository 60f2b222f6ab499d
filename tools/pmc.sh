#!/bin/bash
# tools/pmc.sh <tag> "<counters...>" [bench args]: one --pmc pass over bench.py --profile-pass <bench args> (identical frames, one at a time), prints per-kernel sums
set -u
TAG=$1; CNT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export RPTR_FRAMES_IN_FLIGHT=1
export RPTR_TAIL_BOUNCE=${RPTR_TAIL_BOUNCE:-2}  # every frame of the pass hands over at the same bounce (adaptive would start the first frame without a tail)
cd /tmp; rm -rf /tmp/pmc_$TAG
timeout -k 5 ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $CNT -d /tmp/pmc_$TAG -o $TAG --output-format csv -- python $R/bench.py --profile-pass --steps 3 --warmup 1 "$@" > $OUT/bench.log 2>&1
F=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
[ -z "$F" ] && { echo "no counter file"; tail -5 $OUT/bench.log; exit 1; }
python3 - "$F" "$OUT/summary_$TAG.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); calls=collections.Counter()
seen=set()
for r in rows:
    k=r['Kernel_Name'].split('(')[0]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(r['Dispatch_Id'])
    if key not in seen: seen.add(key); calls[k]+=1
names=sorted({r['Counter_Name'] for r in rows})
with open(sys.argv[2],'w',newline='') as f:
    w=csv.writer(f)  # kernel names contain commas (rp_k_extend<false, true>)
    w.writerow(['kernel','calls']+names)
    for k in sorted(agg, key=lambda k:-calls[k]):
        if not k.startswith(('void rp_k','rp_k')): continue
        w.writerow([k,calls[k]]+['%.6g'%(agg[k][n]) for n in names])
        print('%-50s calls %4d '%(k,calls[k])+' '.join('%s=%.4g'%(n,agg[k][n]) for n in names))
PY
