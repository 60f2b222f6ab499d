#!/bin/bash
# speculative traversal (-DRP_SPEC=1) against the default build under several scheduling thresholds
V=$PWD/gpurun_variants
for a in "" "--scene forest"; do
  for p in "10,48" "4,48" "16,48" "24,48" "4,32" "16,32" "24,32" "32,32" "24,24"; do
    echo "== bench args: $a  preset $p"
    RPTR_TRAVERSE_PRESET=$p BENCH_ARGS="$a" AB_STEPS=60 bash tools/ab.sh $V/lib_base.so $V/lib_spec.so | cut -c1-75
  done
done
