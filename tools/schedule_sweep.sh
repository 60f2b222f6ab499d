#!/bin/bash
# tools/schedule_sweep.sh: frames in flight x frames per launch sequence on C3 / C4 / C5 (the defaults were tuned on C2)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
run() { # label, bench args
  python bench.py $2 --no-cpu-baseline --no-boundary --no-probe --steps ${STEPS:-80} --warmup 5 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('%-34s fif %2d x %d: ms/step %.3f sustained %.3f' % ('$1', d['config']['frames_in_flight'], d['config']['frames_per_launch_sequence'], d['ms_per_step'], (r.get('sustained') or {}).get('ms_per_step', 0)))
"
}
for s in "11 4" "7 4" "16 4" "11 2" "16 2" "6 8"; do set -- $s; run "C4 flattened" "--scene forest --frames-in-flight $1 --batch-frames $2"; done
for s in "11 2" "7 2" "16 2" "11 1" "16 1" "8 4"; do set -- $s; run "C3" "--lights --variant gltf --spp 8 --frames-in-flight $1 --batch-frames $2"; done
for s in "7 1" "5 1" "11 1" "16 1"; do set -- $s; run "C5" "--animate --width 3840 --height 2160 --spp 2 --frames-in-flight $1 --batch-frames $2"; done
