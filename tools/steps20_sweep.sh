# tools/steps20_sweep.sh [tag]: the driver's command (bench.py --gpus 1 --steps 20 --warmup 5: a 26 ms timed region) over launch-sequence length and
# frames in flight -- the pipeline fills and drains inside the region, so the schedule that wins a 200-step run need not win this one
TAG=${1:-r04}; O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
for bf in 1 2 4; do for fif in 3 5 7 11 20; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-boundary --sustained-seconds 0 --batch-frames $bf --frames-in-flight $fif > $O/s20_bf${bf}_fif${fif}_$rep.json 2>/dev/null
done; done; done
python3 - $O <<'PY'
import json,sys,glob,os,collections
acc=collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1]+"/s20_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        acc[os.path.basename(f).rsplit("_",1)[0]].append((d["ms_per_step"], d["config"]["frames_in_flight"]))
    except Exception as e:
        print(f, "unreadable", e)
for k,v in sorted(acc.items(), key=lambda kv: min(x[0] for x in kv[1])):
    print("%-20s contexts %2d  ms/step %s" % (k, v[0][1], " ".join("%.4f" % x[0] for x in v)))
PY
