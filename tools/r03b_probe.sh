mkdir -p gpurun_out/r03b; O=gpurun_out/r03b
timeout 900 python -m pytest tests/test_gpu_device_build.py -x -q -s -k "not c4" > $O/t_device_small.log 2>&1; tail -5 $O/t_device_small.log
timeout 900 python -m pytest tests/test_gpu_device_build.py -x -q -s -k "c4" > $O/t_device_c4.log 2>&1; tail -8 $O/t_device_c4.log
timeout 1500 python -m pytest tests/test_gpu_whole_frames.py -x -q -s > $O/t_whole.log 2>&1; tail -12 $O/t_whole.log
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_validation_cli.py tests/test_abi.py -x -q -m gpu -k "discard or sky or abi" > $O/t_misc.log 2>&1; tail -4 $O/t_misc.log
for cfg in "RPTR_BVH_BUILDER=host" "RPTR_BVH_BUILDER=host RPTR_PRESPLIT=400,1.0" "RPTR_BVH_BUILDER=device"; do
  env $cfg timeout 600 python bench.py --no-cpu-baseline --scene forest --steps 80 > $O/bench_c4_tmp.json 2> $O/bench_c4_tmp.err
  python - "$cfg" $O/bench_c4_tmp.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r=d['roofline']; c=r['counts_per_step']
    print("%-50s ms/step %.3f build %.2fs excl gpu_total %.3f nodes/ray %.2f tris/ray %.2f"%(sys.argv[1],d['ms_per_step'],d['config']['bvh_build_s'],r['stage_ms_per_step']['gpu_total'],c['nodes_closest']/c['rays_closest'],c['tris_closest']/c['rays_closest']))
except Exception as e:
    print(sys.argv[1],"failed",e)
PY
done
