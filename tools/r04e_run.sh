O=gpurun_out/r04e; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
python3 bench.py --no-cpu-baseline > $O/bench_default.json 2>/dev/null
python3 bench.py --no-cpu-baseline --lights --variant gltf --spp 8 > $O/bench_c3.json 2>/dev/null
python3 bench.py --no-cpu-baseline --scene forest --static-camera > $O/bench_c4_static.json 2>/dev/null
python3 bench.py --no-cpu-baseline --animate --width 3840 --height 2160 --spp 2 > $O/bench_c5.json 2>/dev/null
for rep in 1 2 3; do
i=0
for pat in "" "2,4,4,4,4,2" "1,3,4,4,4,4" "1,2,3,4,4,4,2" "3,4,4,4,4,1" "2,3,4,4,4,3" "1,2,4,4,4,4,1"; do
  BENCH_BATCH_PATTERN=$pat python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 > $O/p20_${i}_$rep.json 2>/dev/null
  i=$((i+1))
done; done
python3 - $O <<'PY'
import json,sys,glob,os,collections
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); rf=d["roofline"]
    print(os.path.basename(f), d["ms_per_step"], "sustained", (rf.get("sustained") or {}).get("ms_per_step"), "static", (rf.get("static_camera") or {}).get("ms_per_step"), "stage", rf.get("stage_ms_per_step"), "lat1", rf["latency"]["1"]["ms_per_frame"])
acc=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+"/p20_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    acc[os.path.basename(f).rsplit("_",1)[0]].append((d["ms_per_step"], d["config"]["frames_in_flight"]))
for k,v in sorted(acc.items()):
    print("%-10s contexts %2d  ms/step %s" % (k, v[0][1], " ".join("%.4f" % x[0] for x in v)))
PY
