run() { env $1 python bench.py --no-cpu-baseline --steps ${3:-60} $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-36s %-28s ms/step %.3f latency1 %.3f bvh %s' % ('$1', '$2', d['ms_per_step'], d['roofline']['latency']['1']['ms_per_frame'], d['config']['bvh']))"; }
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "thresholds" 2>&1 | tail -3
for p in 16,32 24,32 32,32 16,24 24,24 20,40 12,32; do run "RPTR_TRAVERSE_PRESET=$p" "--scene forest"; done
run "X=1" "--scene forest --flatten 0"
for p in 0,0 12,48 10,40 14,44; do run "RPTR_TRAVERSE_PRESET=$p" "" 200; done
