run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['one_frame_at_a_time']['stage_ms_per_step']['gpu_total'])"; }
for w in 8 4 2 1; do
  for k in 1 2; do
      echo -n "world $w tail $k: "; RPTR_TAIL_BOUNCE=$k run --emulate-world $w --steps 200
  done
  for t in 65536 262144 1048576; do
      echo -n "world $w adaptive threshold $t: "; RPTR_TAIL_THRESHOLD=$t run --emulate-world $w --steps 200
  done
done
