for rep in 1 2 3; do
for mode in "" "--static-camera"; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --sustained-seconds 0 $mode 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps20 [$mode]', d['ms_per_step'])"
done; done
for b in 0 2 3 4; do
  if [ $b = 0 ]; then unset RPTR_BLOCKS_PER_CU; else export RPTR_BLOCKS_PER_CU=$b; fi
  python3 bench.py --gpus 1 --steps 100 --warmup 5 --no-cpu-baseline --sustained-seconds 0 --frames-in-flight 2 --batch-frames 1 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two frames in flight, blocks per CU $b:', d['ms_per_step'], d['config']['frames_in_flight'], d['config']['frames_per_launch_sequence'])"
done
