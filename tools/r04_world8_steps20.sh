# rank 0's share of an 8-way / 4-way split under the DRIVER's command (--steps 20 --warmup 5): launch-sequence length vs contexts
for w in 8 4; do for rep in 1 2 3; do for bf in 2 4 8; do
  RPTR_MAX_BATCH_SPP=32 RPTR_MAX_BATCH_FRAMES=8 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 --emulate-world $w --batch-frames $bf 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('world $w steps 20 batch $bf:', d['ms_per_step'], 'contexts', d['config']['frames_in_flight'])"
done; done; done
