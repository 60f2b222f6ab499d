# frame time against the tail hand-over threshold for the BASELINE configs
run() { python bench.py --no-cpu-baseline --steps 60 "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'from bounce', d['roofline']['launches_per_step'])"; }
for t in 16384 65536 262144 1048576; do
  echo -n "C3 threshold $t: "; RPTR_TAIL_THRESHOLD=$t run --variant gltf --lights --spp 8
  echo -n "C4 threshold $t: "; RPTR_TAIL_THRESHOLD=$t run --scene forest
  echo -n "C5 threshold $t: "; RPTR_TAIL_THRESHOLD=$t run --animate --width 3840 --height 2160 --spp 2
done
