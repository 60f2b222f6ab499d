#!/bin/bash
# tools/mkvariant.sh <name> <extra hipcc flags...>: builds gpurun_variants/lib_<name>.so (A/B builds for tools/ab.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gpurun_variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -Wno-unused-function "$@" \
  $R/realtimepathtracingresearchframework_amd/csrc/rptr_hip.hip $R/realtimepathtracingresearchframework_amd/csrc/bvh_build.cpp -o $R/gpurun_variants/lib_$N.so
