# tools/r04g_evidence.sh: the round's final evidence on the final build -> gpurun_out/r04g/ (GPU suite, every config's bench line incl. the
# driver's own command, rocprofv3 kernel statistics of the driver's command and of the 200-step run)
O=gpurun_out/r04g; mkdir -p $O
bash tools/round_evidence.sh r04g > $O/evidence.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
bash tools/prof.sh r04g_driver --gpus 1 --steps 20 --warmup 5 > $O/prof_driver_command.txt 2>&1
cp gpurun_out/prof_r04g_driver/*kernel_stats.csv $O/kernel_stats_driver_command_steps20.csv 2>/dev/null
bash tools/prof.sh r04g_exclusive --profile-pass --steps 20 --warmup 2 > $O/prof_exclusive.txt 2>&1
cp gpurun_out/prof_r04g_exclusive/*kernel_stats.csv $O/kernel_stats_exclusive_profile_pass.csv 2>/dev/null
tail -n 22 $O/evidence.txt
python3 - $O <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]+"/bench_driver_command.json").read().strip().splitlines()[-1]); rf=d["roofline"]
print("driver command:", d["ms_per_step"], d["value"], "roofline frac", rf["frac"], "binding", rf.get("binding"), rf.get("binding_frac"), "cpu", d.get("cpu_baseline",{}).get("value"))
for k,v in rf["kernels"].items(): print("  ", k, v.get("launch_ms"), "valu_frac", v.get("valu_frac"), "peak", v.get("valu_peak_ginst_s"), "hbm_frac", v.get("hbm_frac"))
PY
