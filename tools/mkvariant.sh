#!/bin/bash
# tools/mkvariant.sh <name> <extra hipcc flags...>: builds gpurun_variants/lib_<name>.so (A/B builds for tools/ab.sh)
# MKVARIANT_ONLY=<substring>: only the units whose name contains it are compiled with the flags (the rest: the product's objects)
R=$(cd "$(dirname "$0")/.." && pwd)
N=$1; shift
mkdir -p $R/gpurun_variants
cd $R && python3 - "$N" "$@" <<'PY'
import sys, tempfile
from realtimepathtracingresearchframework_amd import build
name, flags = sys.argv[1], sys.argv[2:]
with tempfile.TemporaryDirectory() as d:
    import os
    only = os.environ.get("MKVARIANT_ONLY")
    print(build.build_library(force=True, extra_flags=flags + ["-w"], lib_path="gpurun_variants/lib_%s.so" % name, obj_dir=d,
                              only_units=(lambda n: only in n) if only else None))
PY
