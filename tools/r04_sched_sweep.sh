# pipelined schedule knobs on the no-SLP build: frames in flight x blocks per CU of a traversal launch (C2, 200 steps)
for fif in 7 11 15; do for b in 0 2 3; do
  if [ $b = 0 ]; then unset RPTR_BLOCKS_PER_CU; else export RPTR_BLOCKS_PER_CU=$b; fi
  python3 bench.py --no-cpu-baseline --sustained-seconds 0 --frames-in-flight $fif 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $fif blocks/CU $b:', d['ms_per_step'])"
done; done
