#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   -- kernel-trace stats of bench.py into gpurun_out/prof_<tag>/
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$TAG
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats*") $OUT/ 2>/dev/null
grep '^{' $OUT/bench.log | tail -1
python3 - "$OUT" <<'PY'
import csv,sys,glob
for fn in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for x in csv.DictReader(open(fn)):
        print('%-60s calls %5s total %9.3f ms avg %9.1f us %6s%%' % (x['Name'][:60], x['Calls'], float(x['TotalDurationNs'])/1e6, float(x['AverageNs'])/1e3, x['Percentage']))
PY
python3 - "/tmp/prof_$TAG" <<'PY'
import csv,sys,glob,collections
# per-dispatch durations of the traversal kernels in the last timed step
for fn in glob.glob(sys.argv[1]+'/*kernel_trace.csv'):
    rows=[r for r in csv.DictReader(open(fn))]
    ext=[r for r in rows if 'rp_k_extend<false' in r['Kernel_Name']]
    con=[r for r in rows if 'rp_k_connect<false' in r['Kernel_Name']]
    sh=[r for r in rows if 'rp_k_shade' in r['Kernel_Name']]
    def d(r): return (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    print('extend  last 9 (us):', ' '.join('%.0f'%d(r) for r in ext[-18:-9]))
    print('connect last 9 (us):', ' '.join('%.0f'%d(r) for r in con[-18:-9]))
    print('shade   last 9 (us):', ' '.join('%.0f'%d(r) for r in sh[-18:-9]))
    if ext: print('extend regs: VGPR', ext[-1].get('VGPR_Count'), 'SGPR', ext[-1].get('SGPR_Count'), 'LDS', ext[-1].get('LDS_Block_Size'), 'scratch', ext[-1].get('Scratch_Size'), 'grid', ext[-1].get('Grid_Size'))
PY
