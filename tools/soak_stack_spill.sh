#!/bin/bash
# tools/soak_stack_spill.sh: the traversal's GLOBAL stack path under test. The shipped kernels keep the first RP_LDS_STACK (20) entries of a lane's
# stack in LDS and spill deeper ones to a per-thread slice of global memory; no scene of the suite is deep enough to spill often. This builds the
# library with four LDS entries (gpurun_variants/lib_lds4.so: every query spills, the whole wave takes the generic push / pop path nearly always)
# and runs the bit-exact parity tests on it. Build here (tools/mkvariant.sh lds4 -DRP_LDS_STACK=4), run through gpurun.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
[ -f gpurun_variants/lib_lds4.so ] || tools/mkvariant.sh lds4 -DRP_LDS_STACK=4
RPTR_HIP_LIB=$R/gpurun_variants/lib_lds4.so timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dynamic.py -m gpu -q -x \
  -k "not bench and not ipc and not thresholds and not lds_staged" 2>&1 | grep -E "passed|failed|Error" | tail -3
