#!/bin/bash
# tools/profile_round.sh <tag>: everything profiles/ keeps for one build, into gpurun_out/<tag>/ (run through gpurun from the repository root):
#   the VALU issue-rate microbenchmark, the pmc passes (instructions, cycles, fetch, write, cache) of bench.py --profile-pass for EVERY BASELINE
#   workload -> pmc_<key>_*.csv + pmc_traffic.json (keyed by workload; copy it to profiles/),
#   rocprofv3 --kernel-trace --stats of bench.py --profile-pass (exclusive launches) and of the default bench.py (pipelined),
#   the bench lines of configs[1..4] (C2 default, C3, C4 flattened + two-level, C5), the emulated N-way shares.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bash tools/valu_issue.sh $TAG > $O/valu_issue_run.txt 2>&1
bash tools/pmc_workloads.sh $TAG c2 c3 c3_two_level c4_flat c4_two_level c5 > $O/pmc_workloads.txt 2>&1
bash tools/prof.sh ${TAG}_exclusive --profile-pass --steps 20 --warmup 2 > $O/prof_exclusive.txt 2>&1
cp gpurun_out/prof_${TAG}_exclusive/*kernel_stats.csv $O/kernel_stats_exclusive_profile_pass.csv 2>/dev/null
bash tools/prof.sh ${TAG}_pipelined --steps 200 > $O/prof_pipelined.txt 2>&1
cp gpurun_out/prof_${TAG}_pipelined/*kernel_stats.csv $O/kernel_stats_pipelined_steps200.csv 2>/dev/null
grep '^{' gpurun_out/prof_${TAG}_pipelined/bench.log | tail -1 > $O/bench_under_rocprof_steps200.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --lights --variant gltf --spp 8 > $O/bench_c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --lights --variant gltf --spp 8 --flatten 0 > $O/bench_c3_two_level.json 2>/dev/null
python3 bench.py --scene forest > $O/bench_c4.json 2>/dev/null
RPTR_BVH_BUILDER=host python3 bench.py --no-cpu-baseline --scene forest > $O/bench_c4_host_built_tree.json 2>/dev/null
python3 bench.py --no-cpu-baseline --scene forest --flatten 0 > $O/bench_c4_two_level.json 2>/dev/null
python3 bench.py --animate --width 3840 --height 2160 --spp 2 > $O/bench_c5.json 2>/dev/null
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-world $n > $O/bench_emulated_world$n.json 2>/dev/null; done
# the N > 1 path end to end on this box's one GPU (both ranks on cuda:0: the RCCL probe refuses, the run falls back and says so) with every rank
# under rocprofv3: what tools/prof_ranks.sh <tag> 8 gives on a real node
bash tools/prof_ranks.sh ${TAG}_ranks 2 --same-device > $O/prof_ranks_same_device.txt 2>&1
cp gpurun_out/${TAG}_ranks/bench_gpus2.json $O/bench_gpus2_same_device.json 2>/dev/null
cp gpurun_out/${TAG}_ranks/hbm_gbs_per_rank.txt $O/hbm_gbs_per_rank_same_device.txt 2>/dev/null
python3 - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.JSONDecoder().raw_decode(open(f).read().strip().splitlines()[-1])[0]; rf=d["roofline"]   # (raw_decode: a profiler may append its own words to the line)
        print("%-36s ms/step %.3f Mrays/s %7.0f | excl extend %.3f ms valu_frac %s hbm_counter_frac %s algorithmic bytes over HBM peak %s | frame valu %s | latency 1/2 in flight %s / %s ms | bvh %s" % (
            os.path.basename(f), d["ms_per_step"], d["value"], rf["exclusive_ms_per_step"], rf["valu"]["frac"], rf["hbm_frac"], rf.get("algorithmic_frac_of_hbm_peak"), (rf["valu"]["frame"] or {}).get("pipelined_frac"),
            rf["latency"]["1"]["ms_per_frame"], (rf["latency"].get("2") or {}).get("ms_per_frame"), d["config"].get("bvh", {}).get("built_on")))
    except Exception as e:
        print(f, "unreadable", e)
PY
