#!/bin/bash
# tools/pmc_workloads.sh <tag> [workload keys...]: the five rocprofv3 --pmc passes (instructions, cycles, fetch, write, cache) of
# bench.py --profile-pass for every BASELINE workload -> gpurun_out/<tag>/pmc_<key>_<pass>.{txt,csv} and the entry workloads[<key>] of
# profiles/pmc_traffic.json (what bench.py reads for the hbm / valu figures of that workload's line). Counter passes are separate runs
# with --kernel-trace only (no other trace domain), as the pool requires.
set -u
TAG=${1:-r03}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
KEYS=${@:-c2 c3 c3_two_level c4_flat c4_two_level c5}
args_of() {
  case $1 in
    c2) echo "" ;;
    c3) echo "--lights --variant gltf --spp 8" ;;
    c3_two_level) echo "--lights --variant gltf --spp 8 --flatten 0" ;;
    c4_flat) echo "--scene forest" ;;
    c4_two_level) echo "--scene forest --flatten 0" ;;
    c5) echo "--animate --width 3840 --height 2160 --spp 2" ;;
    *) echo "" ;;
  esac
}
# the bounce at which the tail kernel takes over in the workload's benchmarked schedule (bench.py chooses it adaptively from the previous
# frame's queue lengths; a counter pass renders a handful of frames, so the value is pinned: every frame of the pass then has the same launches)
tail_of() {
  case $1 in
    c2) echo 2 ;;
    c3|c3_two_level|c5) echo 3 ;;
    c4_flat|c4_two_level) echo 5 ;;
    *) echo 2 ;;
  esac
}
for K in $KEYS; do
  A=$(args_of $K)
  export RPTR_TAIL_BOUNCE=$(tail_of $K)
  bash tools/pmc.sh ${K}_insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" $A > $O/pmc_${K}_insts.txt 2>&1
  bash tools/pmc.sh ${K}_cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" $A > $O/pmc_${K}_cycles.txt 2>&1
  bash tools/pmc.sh ${K}_fetch "FETCH_SIZE" $A > $O/pmc_${K}_fetch.txt 2>&1
  bash tools/pmc.sh ${K}_write "WRITE_SIZE" $A > $O/pmc_${K}_write.txt 2>&1
  bash tools/pmc.sh ${K}_tcc "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum" $A > $O/pmc_${K}_tcc.txt 2>&1
  for t in insts cycles fetch write tcc; do cp gpurun_out/pmc_${K}_$t/summary_${K}_$t.csv $O/pmc_${K}_$t.csv 2>/dev/null; done
  python3 tools/make_traffic.py $TAG $K "$A" > $O/pmc_traffic_$K.txt 2>&1
done
cp profiles/pmc_traffic.json $O/pmc_traffic.json
