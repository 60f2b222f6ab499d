mkdir -p gpurun_out/r03d; O=gpurun_out/r03d
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 1 --grid 100x50 --width 320 --height 180 --dist-backend gloo --same-device > $O/two_ranks.out 2> $O/two_ranks.err; echo "rc $?"; grep -v "^\[W\|^W0\|^\*\*\*" $O/two_ranks.err | tail -25; tail -c 400 $O/two_ranks.out
for sc in 0 1; do for cfg in "" "--lights --variant gltf --spp 8"; do RPTR_SIDE_CONNECT=$sc python bench.py --no-cpu-baseline --steps 40 $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_connect $sc [$cfg]: ms/step', d['ms_per_step'], 'latency1', d['roofline']['latency']['1']['ms_per_frame'], 'latency2', d['roofline']['latency']['2']['ms_per_frame'])"; done; done
for sc in 0 1; do RPTR_SIDE_CONNECT=$sc python bench.py --no-cpu-baseline --steps 40 --emulate-world 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('side_connect $sc [1/8 frame]: ms/step', d['ms_per_step'], 'latency1', d['roofline']['latency']['1'])"; done
