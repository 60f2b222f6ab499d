mkdir -p gpurun_out/r03c; O=gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_device_build.py -x -q -s > $O/t_device.log 2>&1; tail -8 $O/t_device.log
timeout 900 python -m pytest tests/test_gpu_whole_frames.py -x -q -s -k "c5" > $O/t_whole_c5.log 2>&1; tail -4 $O/t_whole_c5.log
timeout 900 python -m pytest tests/test_validation_cli.py tests/test_gpu_parity.py -x -q -s -k "sky or dry_run or two_ranks or starts_its_own" > $O/t_misc.log 2>&1; tail -6 $O/t_misc.log
for cfg in "RPTR_BVH_BUILDER=host" "RPTR_BVH_BUILDER=device"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --scene forest --steps 80 > $O/bench_c4_tmp.json 2> $O/bench_c4_tmp.err
  python - "$cfg" $O/bench_c4_tmp.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d['roofline']; c=r['counts_per_step']
    print("%-50s ms/step %.3f build %.2fs excl gpu_total %.3f nodes/ray %.2f tris/ray %.2f latency %s bvh %s"%(sys.argv[1],d['ms_per_step'],d['config']['bvh_build_s'],r['stage_ms_per_step']['gpu_total'],c['nodes_closest']/c['rays_closest'],c['tris_closest']/c['rays_closest'],r['latency'],d['config']['bvh']))
except Exception as e:
    print(sys.argv[1],"failed",e); print(open(sys.argv[2].replace('.json','.err')).read()[-1500:])
PY
done
for st in 20 200; do for bf in 2 4; do python bench.py --no-cpu-baseline --steps $st --batch-frames $bf 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 steps $st batch $bf: ms/step', d['ms_per_step'], 'fif', d['config']['frames_in_flight'], 'latency', d['roofline']['latency'], 'binding', d['roofline']['binding_frac'], 'frac', d['roofline']['frac'])"; done; done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
