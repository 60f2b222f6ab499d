#!/bin/bash
# tools/profile_round.sh <tag>: everything profiles/ keeps for one build, into gpurun_out/<tag>/ (run through gpurun from the repository root):
#   pmc passes (instructions, cycles, fetch, write, cache) of bench.py --profile-pass -> pmc_*.csv + pmc_traffic.json (also written to profiles/),
#   rocprofv3 --kernel-trace --stats of bench.py --profile-pass (exclusive launches) and of the default bench.py (pipelined),
#   the bench lines of configs[1..4] (C2 default, C3, C4 flattened + two-level, C5), the emulated N-way shares.
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
bash tools/pmc.sh insts "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" > $O/pmc_insts.txt 2>&1
bash tools/pmc.sh cycles "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" > $O/pmc_cycles.txt 2>&1
bash tools/pmc.sh fetch "FETCH_SIZE" > $O/pmc_fetch.txt 2>&1
bash tools/pmc.sh write "WRITE_SIZE" > $O/pmc_write.txt 2>&1
bash tools/pmc.sh tcc "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum" > $O/pmc_tcc.txt 2>&1
for t in insts cycles fetch write tcc; do cp gpurun_out/pmc_$t/summary_$t.csv $O/pmc_$t.csv 2>/dev/null; done
python3 tools/make_traffic.py $TAG > $O/pmc_traffic.txt 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
bash tools/prof.sh ${TAG}_exclusive --profile-pass --steps 20 --warmup 2 > $O/prof_exclusive.txt 2>&1
cp gpurun_out/prof_${TAG}_exclusive/*kernel_stats.csv $O/kernel_stats_exclusive_profile_pass.csv 2>/dev/null
bash tools/prof.sh ${TAG}_pipelined --steps 200 > $O/prof_pipelined.txt 2>&1
cp gpurun_out/prof_${TAG}_pipelined/*kernel_stats.csv $O/kernel_stats_pipelined_steps200.csv 2>/dev/null
grep '^{' gpurun_out/prof_${TAG}_pipelined/bench.log | tail -1 > $O/bench_under_rocprof_steps200.json
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --no-cpu-baseline --lights --variant gltf --spp 8 > $O/bench_c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --scene forest > $O/bench_c4.json 2>/dev/null
python3 bench.py --no-cpu-baseline --scene forest --flatten 0 > $O/bench_c4_two_level.json 2>/dev/null
python3 bench.py --no-cpu-baseline --animate --width 3840 --height 2160 --spp 2 > $O/bench_c5.json 2>/dev/null
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-world $n > $O/bench_emulated_world$n.json 2>/dev/null; done
python3 - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); rf=d["roofline"]
        print("%-36s ms/step %.3f Mrays/s %7.0f | excl extend %.3f ms valu_frac %s hbm_frac %s alg_frac %s | frame valu %s" % (os.path.basename(f), d["ms_per_step"], d["value"], rf["exclusive_ms_per_step"], rf["valu"]["frac"], rf["hbm_frac"], rf["algorithmic_frac"], (rf["valu"]["frame"] or {}).get("pipelined_frac")))
    except Exception as e:
        print(f, "unreadable", e)
PY
