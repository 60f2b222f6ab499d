run() { env $1 python bench.py --no-cpu-baseline --steps ${3:-200} $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['roofline']['stage_ms_per_step']; print('%-58s %-40s ms/step %.3f latency1 %.3f | excl extend %.3f connect %.3f' % ('$1', '$2', d['ms_per_step'], d['roofline']['latency']['1']['ms_per_frame'], s['extend'], s['connect']))"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lds_staged" 2>&1 | tail -3
run "RPTR_LDS_TOP=0" ""
run "RPTR_LDS_TOP=1" ""
run "RPTR_LDS_TOP=1 RPTR_HIP_LIB=gpurun_variants/lib_top256.so" ""
run "RPTR_LDS_TOP=0" ""
run "RPTR_LDS_TOP=0" "--animate --width 3840 --height 2160 --spp 2" 60
run "RPTR_LDS_TOP=1" "--animate --width 3840 --height 2160 --spp 2" 60
run "RPTR_LDS_TOP=0" "--scene forest" 60
run "RPTR_LDS_TOP=1" "--scene forest" 60
run "RPTR_LDS_TOP=1 RPTR_HIP_LIB=gpurun_variants/lib_top256.so" "--scene forest" 60
