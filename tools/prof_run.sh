mkdir -p gpurun_out/ab
for cfg in "--scene forest --flatten 0" "--scene forest --flatten 1" ""; do
  echo "=== bench.py $cfg (RP_PROF build, one frame at a time, static camera)"
  RPTR_HIP_LIB=gpurun_variants/lib_prof.so python bench.py $cfg --frames-in-flight 1 --batch-frames 1 --steps 2 --warmup 1 --static-camera --no-cpu-baseline --no-boundary --no-probe 2>&1 | grep -E "RP_PROF|^\{" | tail -12 | cut -c1-900
done
