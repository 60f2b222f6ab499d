#!/bin/bash
# tools/fetch_sweep.sh: option traverse_fetch (pool size of a traversal wave) against the number of frames in flight
for f in ${FETCHES:-0 384 512 768}; do
RPTR_TRAVERSE_FETCH=$f python bench.py --no-cpu-baseline --no-probe --steps 100 --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; L=r['latency']; b=d.get('boundary') or {}
        print('traverse_fetch %4s: pipelined %.3f sustained %.3f | one at a time %.3f | two in flight %.3f | boundary swap_buffers_2 %s full %s' % ('$f', d['ms_per_step'], (r.get('sustained') or {}).get('ms_per_step', 0), L['1']['ms_per_frame'], (L.get('2') or {}).get('ms_per_frame', 0), (b.get('swap_buffers_2') or {}).get('ms_per_frame'), (b.get('full_schedule') or {}).get('ms_per_frame')))
"
done
