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
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o $TAG --output-format csv -- python $R/bench.py --no-cpu-baseline "$@" > $OUT/bench.log 2>&1
cp $(find /tmp/prof_$TAG -name "*kernel_stats*") $OUT/ 2>/dev/null
grep '^{' $OUT/bench.log | tail -1
python3 - "$OUT" <<'PY'
import csv,sys,glob
for fn in glob.glob(sys.argv[1]+'/*kernel_stats.csv'):
    for x in csv.DictReader(open(fn)):
        print('%-60s calls %5s total %9.3f ms avg %9.1f us %6s%%' % (x['Name'][:60], x['Calls'], float(x['TotalDurationNs'])/1e6, float(x['AverageNs'])/1e3, x['Percentage']))
PY
