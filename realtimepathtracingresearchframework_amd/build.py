"""Builds librptr_hip.so (the C-ABI shared library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build
container; the resulting .so travels with the repository snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "librptr_hip.so")
# Translation units: (object name, source, extra flags). The path stages are templates with a few hundred instantiations between them;
# one family per unit (csrc/launch.h), the units compile side by side.
UNITS = [
    ("rptr_hip", "rptr_hip.hip", []),
    ("bvh_build", "bvh_build.cpp", []),
    ("k_extend", "k_extend.hip", []),
] + [("k_shade_v%d%s" % (v, "_fast" if m else ""), "k_shade.hip", ["-DRP_INST_VARIANT=%d" % v, "-DRP_FAST_MATH=%d" % m]) for m in range(2) for v in range(3)] \
  + [("k_tail_v%d%s" % (v, "_fast" if m else ""), "k_tail.hip", ["-DRP_INST_VARIANT=%d" % v, "-DRP_FAST_MATH=%d" % m]) for m in range(2) for v in range(3)]
# (k_shade / k_tail: once per gpu-program variant and per build of the shading arithmetic -- IEEE division / square root, the oracle's bits,
# or the hardware's 1-ulp reciprocal / square root: option "fast_math", csrc/dmath.h)
SOURCES = sorted({u[1] for u in UNITS})
HEADERS = ["kernels.h", "kernels_misc.h", "launch.h", "host_state.h", "host_bvh.inl", "host_scene.inl", "host_frame.inl", "host_access.inl", "host_comm.h", "lbvh.h", "ploc.h", "dtraverse.h", "bvh4.h", "dshade.h", "dmath.h", "bvh_build.h",
           "../../include/rptr_hip.h", "../../include/rptr_bvh.h"]
OBJ_DIR = os.path.join(CSRC, "obj")

# -ffp-contract=off: fused multiply-adds only where the reference writes fma()
# itself; keeps images bit-reproducible across launches/tilings and comparable
# with the CPU oracle (DESIGN.md "Numerics").
# -fno-slp-vectorize (round 4): left alone, the SLP vectoriser pairs the shading code's scalar float operations into v_pk_mul_f32 /
# v_pk_add_f32 -- which issue at HALF rate on gfx950 (tools/microbench/valu_issue.hip: two flops per slot either way) -- and pays for
# it with v_mov_b32 to pack and unpack their operands and with register pairs: glTF shade kernel 552 -> 326 moves, 438 -> 0 packed
# operations, no scratch; closest-hit kernels 95 / 90 -> 77 / 76 VGPRs. Same IEEE operations on the same values (the explicitly packed
# slab test of dtraverse.h is untouched: it uses vector types): images bit-identical. Pipelined frames -4...-6 % on every configuration
# (profiles/r04_notes.md section 9).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP backend cannot be built (there is no CPU fallback)")


def source_id():
    """sha256 (16 hex digits) over everything the library is compiled from -- sources, headers, flags: what rptr_hip_build_id() returns.
    profiles/pmc_traffic.json records the id of the library its counter passes ran on; bench.py flags counters of another build as stale."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(set(SOURCES + HEADERS)):
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr([(n, fl) for n, _, fl in UNITS]).encode())
    return h.hexdigest()[:16]


def _newest_input():
    return max(os.path.getmtime(os.path.join(CSRC, f)) for f in SOURCES + HEADERS)


def needs_build(lib_path=LIB_PATH):
    return not os.path.exists(lib_path) or _newest_input() > os.path.getmtime(lib_path)


def build_library(force=False, verbose=False, extra_flags=(), lib_path=LIB_PATH, obj_dir=None, jobs=None, only_units=None):
    """hipcc -c per unit (in parallel), then one link. extra_flags / lib_path / obj_dir: measurement builds (tools/mkvariant.sh);
    only_units (a predicate on the unit name): compile just those with the extra flags and link them with the product's other objects."""
    if not force and not needs_build(lib_path):
        return lib_path
    from concurrent.futures import ThreadPoolExecutor
    obj_dir = obj_dir or OBJ_DIR
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    build_id = source_id()

    def compile_unit(unit):
        name, src, flags = unit
        if only_units is not None and not only_units(name):
            return os.path.join(OBJ_DIR, name + ".o"), 0, ""
        obj = os.path.join(obj_dir, name + ".o")
        cmd = [hipcc] + FLAGS + list(extra_flags) + flags + (['-DRP_BUILD_ID="%s%s"' % (build_id, "+" + "".join(extra_flags) if extra_flags else "")] if name == "rptr_hip" else []) \
            + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
        return obj, r.returncode, r.stdout

    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_unit, UNITS))
    for obj, rc, out in results:
        if out.strip():
            print(out)
        if rc != 0:
            raise RuntimeError("hipcc failed for %s" % obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [r[0] for r in results] + ["-o", lib_path]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


HOST_TOOLS = {"rptr_hip": "rptr_cli.cpp", "demo_host": "demo_host.cpp"}
PLAIN_TOOLS = {"rptr_compare": ("rptr_compare.cpp", ["-lz"])}  # no backend behind them
BIN_DIR = os.path.join(HERE, "bin")


def build_host_tools(verbose=False):
    """The C++ host programs over the C ABI (g++, no HIP headers): bin/rptr_hip (the reference's headless --validation / --profiling runs:
    .pfm images, profiling CSV), bin/demo_host, bin/rptr_compare (the reference's compare_exr)."""
    os.makedirs(BIN_DIR, exist_ok=True)
    out = []
    for name, src in HOST_TOOLS.items():
        exe = os.path.join(BIN_DIR, name)
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", os.path.join(HERE, "host", src), "-o", exe, "-L" + HERE, "-lrptr_hip",
               "-Wl,-rpath," + HERE, "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        out.append(exe)
    for name, (src, libs) in PLAIN_TOOLS.items():
        exe = os.path.join(BIN_DIR, name)
        cmd = ["g++", "-std=c++17", "-O2", "-Wall", os.path.join(HERE, "host", src), "-o", exe] + libs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        out.append(exe)
    return out


if __name__ == "__main__":
    build_library(force=True, verbose=True)
    build_host_tools(verbose=True)
