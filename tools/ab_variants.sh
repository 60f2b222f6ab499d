for lib in gpurun_variants/lib_base.so gpurun_variants/lib_nm8.so gpurun_variants/lib_nm13.so gpurun_variants/lib_nm16.so gpurun_variants/lib_rf40.so gpurun_variants/lib_rf56.so gpurun_variants/lib_lds16.so gpurun_variants/lib_fd2.so gpurun_variants/lib_base.so; do
  a=$(RPTR_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['one_frame_at_a_time']['stage_ms_per_step']['extend'], d['roofline']['one_frame_at_a_time']['stage_ms_per_step']['connect'])")
  b=$(RPTR_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 40 --variant gltf --lights --spp 8 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$(basename $lib): C2 $a | C3 $b"
done
