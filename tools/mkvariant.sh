#!/bin/bash
# tools/mkvariant.sh <name> <extra hipcc flags...>: builds gpurun_variants/lib_<name>.so (A/B builds for tools/ab.sh)
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gpurun_variants
cd $R && python3 - "$N" "$@" <<'PY'
import sys, tempfile
from realtimepathtracingresearchframework_amd import build
name, flags = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    print(build.build_library(force=True, extra_flags=flags + ["-w"], lib_path="gpurun_variants/lib_%s.so" % name, obj_dir=d))
PY
