#!/bin/bash
# tools/validation_timing.sh [spp] [batch_spp]: bin/rptr_hip --validation on C2 (1 M triangles, 1080p, Lambert) -- the queued accumulation
# (launch sequences of up to 16 samples, two in flight) against --synchronous frames: wall time of both, and the two PFMs compared byte for byte
# ("C2 64 spp wall time within 10 % of 16 x ms_per_step, identical PFM bits")
set -u
SPP=${1:-64}; B=${2:-4}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$(mktemp -d)
cd $R
python3 -c "
from realtimepathtracingresearchframework_amd import scenes
scenes.grid_1m().dump('$T/c2.rpsc')
"
EXE=realtimepathtracingresearchframework_amd/bin/rptr_hip
for mode in queued synchronous; do
  extra=""; [ $mode = synchronous ] && extra="--synchronous"
  for rep in 1 2; do
    $EXE $T/c2.rpsc --validation $T/$mode --validation-spp $SPP --batch-spp $B --img 1920 1080 --variant diffuse --pfm $extra | grep -E "wall|spp in" | tr '\n' ' '
    echo " [$mode, run $rep]"
  done
done
cmp $T/queued_$(printf %04d $SPP).pfm $T/synchronous_$(printf %04d $SPP).pfm && echo "PFM bits identical ($SPP spp, frames of $B samples)"
rm -rf $T
