# tools/round_evidence.sh [tag]: the GPU suite, then the bench lines of every config on the current build -> gpurun_out/<tag>/ (profiles/r03m_* came from it)
TAG=${1:-r03m}; O=gpurun_out/$TAG; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_default_steps20.json 2>/dev/null
python3 bench.py --lights --variant gltf --spp 8 > $O/bench_c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --lights --variant gltf --spp 8 --flatten 0 > $O/bench_c3_two_level.json 2>/dev/null
python3 bench.py --scene forest > $O/bench_c4.json 2>/dev/null
RPTR_BVH_BUILDER=host python3 bench.py --no-cpu-baseline --scene forest > $O/bench_c4_host_built_tree.json 2>/dev/null
python3 bench.py --no-cpu-baseline --scene forest --flatten 0 > $O/bench_c4_two_level.json 2>/dev/null
python3 bench.py --animate --width 3840 --height 2160 --spp 2 > $O/bench_c5.json 2>/dev/null
for n in 2 4 8; do python3 bench.py --no-cpu-baseline --emulate-world $n > $O/bench_emulated_world$n.json 2>/dev/null; done
python3 bench.py --gpus 2 --same-device --steps 40 --no-cpu-baseline > $O/bench_gpus2_same_device.json 2> $O/bench_gpus2_same_device.err
python3 bench.py --gpus 2 --same-device --steps 40 --no-cpu-baseline --gather ipc > $O/bench_gpus2_same_device_ipc_gather.json 2> $O/bench_gpus2_same_device_ipc_gather.err
bash tools/prof.sh ${TAG}_pipelined --steps 200 > $O/prof_pipelined.txt 2>&1
cp gpurun_out/prof_${TAG}_pipelined/*kernel_stats.csv $O/kernel_stats_pipelined_steps200.csv 2>/dev/null
python3 - $O <<'PY'
import json,sys,glob,os
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.JSONDecoder().raw_decode(open(f).read().strip().splitlines()[-1])[0]; rf=d["roofline"]   # (raw_decode: a profiler may append its own words to the line)
        print("%-36s ms/step %.3f Mrays/s %7.0f | excl extend %.3f ms valu_frac %s hbm_counter_frac %s algorithmic bytes over HBM peak %s | frame valu %s | latency 1/2 in flight %s / %s ms | bvh %s %s" % (
            os.path.basename(f), d["ms_per_step"], d["value"], rf["exclusive_ms_per_step"], rf["valu"]["frac"], rf["hbm_frac"], rf.get("algorithmic_frac_of_hbm_peak"), (rf["valu"]["frame"] or {}).get("pipelined_frac"),
            rf["latency"]["1"]["ms_per_frame"], (rf["latency"].get("2") or {}).get("ms_per_frame"), d["config"].get("bvh", {}).get("built_on"), d["config"].get("bvh", {}).get("traversal_thresholds")))
    except Exception as e:
        print(f, "unreadable", e)
PY
