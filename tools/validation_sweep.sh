#!/bin/bash
# tools/validation_sweep.sh: bin/rptr_hip --validation, 64 spp of C2 in frames of 4 samples: frame contexts x frames per launch sequence x the
# tail kernel's first hand-over (experiment behind profiles/r05_notes.md section 9)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d)
cd $R
python3 -c "
from realtimepathtracingresearchframework_amd import scenes
scenes.grid_1m().dump('$T/c2.rpsc')
"
EXE=realtimepathtracingresearchframework_amd/bin/rptr_hip
run() { # label, env, flags
  for rep in 1 2; do
    echo -n "$1 [run $rep]: "
    env $2 $EXE $T/c2.rpsc --validation $T/v --validation-spp 64 --batch-spp 4 --img 1920 1080 --variant diffuse --pfm $3 | grep -E "wall" | sed 's/.*: wall/wall/'
  done
  md5sum $T/v_0064.pfm | cut -c1-12
}
run "4 contexts x 4 frames (default)" "A=1" ""
run "4 x 4, tail kernel from bounce 2 at once" "RPTR_TAIL_BOUNCE=2" ""
run "8 x 2" "A=1" "--frames-in-flight 8 --frames-per-launch 2"
run "8 x 2, tail 2" "RPTR_TAIL_BOUNCE=2" "--frames-in-flight 8 --frames-per-launch 2"
run "16 x 1, tail 2" "RPTR_TAIL_BOUNCE=2" "--frames-in-flight 16 --frames-per-launch 1"
run "6 x 3, tail 2" "RPTR_TAIL_BOUNCE=2" "--frames-in-flight 6 --frames-per-launch 3"
run "2 x 8, tail 2" "RPTR_TAIL_BOUNCE=2" "--frames-in-flight 2 --frames-per-launch 8"
rm -rf $T
