export RPTR_BVH_BUILDER=host
echo "== C4 variants"; BENCH_ARGS="--scene forest" AB_STEPS=60 bash tools/ab.sh gpurun_variants/lib_base.so gpurun_variants/lib_r24.so gpurun_variants/lib_r32.so gpurun_variants/lib_r40.so gpurun_variants/lib_r32n16.so gpurun_variants/lib_r32f48.so gpurun_variants/lib_base.so
echo "== C2 variants"; AB_STEPS=200 bash tools/ab.sh gpurun_variants/lib_base.so gpurun_variants/lib_r24.so gpurun_variants/lib_r32.so gpurun_variants/lib_r40.so gpurun_variants/lib_r32n16.so gpurun_variants/lib_r32f48.so gpurun_variants/lib_base.so
echo "== C3 variants"; BENCH_ARGS="--lights --variant gltf --spp 8" AB_STEPS=60 bash tools/ab.sh gpurun_variants/lib_base.so gpurun_variants/lib_r32.so gpurun_variants/lib_r40.so gpurun_variants/lib_r32f48.so
echo "== C5 variants"; BENCH_ARGS="--animate --width 3840 --height 2160 --spp 2" AB_STEPS=60 bash tools/ab.sh gpurun_variants/lib_base.so gpurun_variants/lib_r32.so gpurun_variants/lib_r40.so gpurun_variants/lib_r32f48.so
