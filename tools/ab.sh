#!/bin/bash
# A/B of library variants: tools/ab.sh <lib.so> [<lib.so> ...]  (bench.py, no CPU baseline), prints ms/frame + stage split
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in "$@"; do
  RPTR_HIP_LIB=$lib python $R/bench.py --no-cpu-baseline --steps ${AB_STEPS:-100} --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d['roofline']['stage_ms_per_step']
        print('%-40s ms/step %.3f  Mrays/s %.0f | one frame at a time: total %.3f extend %.3f connect %.3f shade %.3f tail %.3f resolve %.3f | latency 1 / 2 in flight %.3f / %s' % ('$(basename $lib)', d['ms_per_step'], d['value'], s['gpu_total'], s['extend'], s['connect'], s['shade'], s['tail'], s['resolve'], d['roofline']['latency']['1']['ms_per_frame'], (d['roofline']['latency'].get('2') or {}).get('ms_per_frame')))
"
done
