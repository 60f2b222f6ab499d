"""CPU checks of the drop-in boundary: struct layouts, exported symbols, loud failure without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from realtimepathtracingresearchframework_amd import abi, backend, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_sizes_match_reference_layouts():
    # sizes quoted in SURVEY 7.1-M1 / 8(a)
    assert C.sizeof(abi.BaseMaterial) == 80          # rendering/bsdfs/base_material.h.glsl:13-34
    assert C.sizeof(abi.TriLightData) == 48          # rendering/lights/tri.h.glsl:13-26
    assert C.sizeof(abi.RenderRayQuery) == 32        # librender/render_params.glsl.h:165-170
    assert C.sizeof(abi.RenderParams) == 80          # librender/render_params.glsl.h:130-155
    assert C.sizeof(abi.SkyModelParams) == 160       # sky_model.h.glsl:7-10
    assert abi.BaseMaterial.emission_intensity.offset == 76
    assert abi.BaseMaterial.ior.offset == 48
    assert abi.RenderRayQuery.t_max.offset == 28


def test_header_declares_exactly_the_exported_symbols():
    text = open(os.path.join(ROOT, "include", "rptr_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(rptr_hip_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(abi.EXPORTED_SYMBOLS)


def test_library_loads_and_exports_every_symbol():
    if not os.path.exists(build.LIB_PATH):
        build.build_library()
    L = backend.load_library()
    for name in abi.EXPORTED_SYMBOLS:
        assert hasattr(L, name), name
    assert b"HIP" in L.rptr_hip_name()


def test_abi_version_is_checked_before_anything_else():
    """RptrCreateInfo.abi_version (struct padding was given a meaning without a guard): the header's number = the library's =
    abi.py's, and a caller compiled against another layout -- or one that left the former padding word at 0 -- is refused by
    rptr_hip_create before any device work, GPU or not"""
    hdr = open(os.path.join(ROOT, "include", "rptr_hip.h")).read()
    version = int(re.search(r"#define RPTR_HIP_ABI_VERSION\s+(\d+)", hdr).group(1))
    L = backend.load_library()
    L.rptr_hip_abi_version.restype = C.c_int
    assert L.rptr_hip_abi_version() == version == abi.ABI_VERSION
    assert abi.CreateInfo.abi_version.offset == 28 and abi.CreateInfo.flags.offset == 32 and C.sizeof(abi.CreateInfo) == 40
    for bad in (0, version - 1, version + 1):
        info = abi.CreateInfo(0, 0, 1, 32, None, 1, bad, 0, 0)
        h = C.c_void_p()
        assert L.rptr_hip_create(C.byref(info), C.byref(h)) == abi.RPTR_E_INVALID and not h.value
        assert b"abi_version" in L.rptr_hip_last_error(None)
    # RptrTextureDesc.mip_levels used to be padding: the field is where the header says, and the mirror names it
    assert abi.TextureDesc.mip_levels.offset == 20 and C.sizeof(abi.TextureDesc) == 24


def test_options_are_part_of_the_c_abi_and_the_environment_only_overrides():
    """include/rptr_hip.h "Options" (what decides the benchmarked numbers was read with getenv and absent from the ABI): every
    switch is a named integer reachable through rptr_hip_set_option / _get_option -- handle-less here, as the process default --, unknown
    keys and values out of range are refused, the header's table names every key, and an option's environment variable wins over the
    default. The default of "flatten" is auto: rptr_hip_build_bvh_host (what set_scene does, on the host) flattens a static multi-instance
    scene without being told to."""
    import subprocess
    import sys
    L = backend.load_library()
    hdr = open(os.path.join(ROOT, "include", "rptr_hip.h")).read()
    n = L.rptr_hip_option_count()
    keys = [L.rptr_hip_option_name(i).decode() for i in range(n)]
    assert n >= 20 and L.rptr_hip_option_name(n) is None and len(set(keys)) == n
    for k in keys:
        assert re.search(r"\b%s\b" % k, hdr), "include/rptr_hip.h does not document option %s" % k
    v = C.c_int64(7)
    assert L.rptr_hip_get_option(None, b"flatten", C.byref(v)) == 0 and v.value == int(os.environ.get("RPTR_FLATTEN", "-1"))
    assert L.rptr_hip_get_option(None, b"max_batch_frames", C.byref(v)) == 0 and v.value == 8
    assert L.rptr_hip_set_option(None, b"no_such_option", 1) == abi.RPTR_E_INVALID and b"no_such_option" in L.rptr_hip_last_error(None)
    assert L.rptr_hip_set_option(None, b"flatten", 5) == abi.RPTR_E_INVALID
    assert L.rptr_hip_get_option(None, b"nope", C.byref(v)) == abi.RPTR_E_INVALID
    # a fresh process: default auto-flatten, the process default switched off through the ABI, and the environment's last word
    probe = ("import sys; sys.path.insert(0, %r)\n"
             "import numpy as np\n"
             "from realtimepathtracingresearchframework_amd import backend, scenes\n"
             "L = backend.load_library(); s = scenes.two_level_test()\n"
             "def flat():\n"
             "    return int((np.frombuffer(np.ascontiguousarray(backend.build_bvh_host(s)[1]).tobytes(), np.uint32).reshape(-1, 12)[:, 11] >> 8).max() > 0)\n"
             "a = flat(); L.rptr_hip_set_option(None, b'flatten', 0); b = flat(); L.rptr_hip_set_option(None, b'flatten', -1); c = flat()\n"
             "print(a, b, c)") % ROOT
    env = {k: v for k, v in os.environ.items() if not k.startswith("RPTR_")}
    env["RPTR_SKIP_TORCH_PRELOAD"] = "1"
    out = subprocess.run([sys.executable, "-c", probe], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["1", "0", "1"]
    out = subprocess.run([sys.executable, "-c", probe], env=dict(env, RPTR_FLATTEN="0"), capture_output=True, text=True, check=True).stdout.split()
    assert out == ["0", "0", "0"]


def test_hardware_queues_are_the_hosts_unless_the_create_info_says_otherwise():
    """GPU_MAX_HW_QUEUES (one hardware queue per frame context's stream: DESIGN.md section 3) belongs to the host process. Loading the library
    changes nothing (round 5 set it from a load-time constructor: VERDICT r5 weak 11); a create WITHOUT RPTR_CREATE_SET_HW_QUEUES changes
    nothing either; a create WITH the flag -- what bin/rptr_hip passes -- sets it before its first HIP call when nobody did, and leaves a
    value the host chose alone. (The create itself may fail here for want of a GPU: the variable is set before the device is looked for.)"""
    import subprocess
    import sys
    # (os.environ is a snapshot: ask the C library)
    probe = ("import os, sys, ctypes as C\nsys.path.insert(0, %r)\n"
             "from realtimepathtracingresearchframework_amd import abi\n"
             "L = C.CDLL(%r)\ng = C.CDLL(None).getenv\ng.restype = C.c_char_p\n"
             "out = [(g(b'GPU_MAX_HW_QUEUES') or b'unset').decode()]\n"
             "info = abi.CreateInfo(0, 0, 1, 32, None, 11, abi.ABI_VERSION, int(sys.argv[1]), 0); h = C.c_void_p()\n"
             "L.rptr_hip_create(C.byref(info), C.byref(h))\n"
             "out.append((g(b'GPU_MAX_HW_QUEUES') or b'unset').decode())\n"
             "print(' '.join(out))") % (ROOT, build.LIB_PATH)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    env["RPTR_QUIET"] = "1"

    def run(flags, **extra):
        return subprocess.run([sys.executable, "-c", probe, str(flags)], env=dict(env, **extra), capture_output=True, text=True, check=True).stdout.split()
    assert run(0) == ["unset", "unset"]                                  # loading and creating: the host's environment is the host's
    out = run(abi.CREATE_SET_HW_QUEUES)
    assert out[0] == "unset" and int(out[1]) >= 13, out                  # the process's first create, with the host's say-so
    assert run(abi.CREATE_SET_HW_QUEUES, GPU_MAX_HW_QUEUES="6") == ["6", "6"]   # a value the host chose stays


def test_no_cpu_fallback_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(backend.BackendError) as e:
        backend.RenderHip()
    assert e.value.code == abi.RPTR_E_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "realtimepathtracingresearchframework_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in src and "oracle_lib" not in src and "orc_" not in src, os.path.join(dirpath, f)


ABI_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "libabi_ref.so")


@pytest.mark.skipif(not os.path.exists(ABI_REF), reason="oracle/_ref/libabi_ref.so is built from the reference tree (oracle/Makefile)")
def test_boundary_structs_and_constants_against_the_references_headers():
    """pinned by oracle/_ref: the reference's dual-language headers (render_params.glsl.h, base_material.h.glsl, tri.h.glsl,
    sky_model.h.glsl, pathspace.h, sobol_data.h, bn_data.h, texture_channel_mask.h) compiled where they lie report sizes, member
    offsets, default values and constants; the ctypes mirrors of include/rptr_hip.h, their defaults and the header's constants agree"""
    import json
    import struct
    L = C.CDLL(ABI_REF)
    L.ref_abi_json.restype = C.c_char_p
    ref = json.loads(L.ref_abi_json())
    for cls, name in ((abi.RenderParams, "RenderParams"), (abi.LightSamplingConfig, "LightSamplingConfig"), (abi.BaseMaterial, "BaseMaterial"),
                      (abi.RenderRayQuery, "RenderRayQuery"), (abi.TriLightData, "TriLightData"), (abi.SkyModelParams, "SkyModelParams")):
        assert C.sizeof(cls) == ref["sizeof_" + name], name
        alias = {"v0_x": "v0", "v1_x": "v1", "v2_x": "v2", "radiance_x": "radiance"}
        seen = 0
        for key, off in ref.items():
            if key.startswith("offsetof_%s_" % name):
                field = key[len("offsetof_%s_" % name):]
                assert getattr(cls, alias.get(field, field)).offset == off, key
                seen += 1
        assert seen >= 2, name
    for obj, name in ((abi.RenderParams.default(), "RenderParams"), (abi.LightSamplingConfig.default(), "LightSamplingConfig"),
                      (abi.make_material(flags=0), "BaseMaterial")):
        n = 0
        for key, val in ref.items():
            if key.startswith("default_%s_" % name) and not key.endswith(("_x", "_y")):
                assert np.float32(getattr(obj, key[len("default_%s_" % name):])) == np.float32(val), key
                n += 1
        assert n >= 4, name
    m = abi.make_material(flags=0)
    assert np.float32(m.base_color[0]) == np.float32(ref["default_BaseMaterial_base_color_x"]) and m.transmission_color[0] == ref["default_BaseMaterial_transmission_color_x"]
    # constants of the header (through abi.py, which test_struct_sizes / the symbol tests hold against include/rptr_hip.h)
    hdr = open(os.path.join(os.path.dirname(ABI_REF), "..", "..", "include", "rptr_hip.h")).read()
    def define(name):
        return eval(re.search(r"#define %s\s+(.+?)\s*(/\*|$)" % name, hdr, re.M).group(1).replace("u", "").replace("f", ""))
    assert define("RPTR_MAX_PATH_DEPTH") == abi.MAX_PATH_DEPTH == ref["MAX_PATH_DEPTH"]
    assert define("RPTR_DEFAULT_RR_PATH_DEPTH") == abi.DEFAULT_RR_PATH_DEPTH == ref["DEFAULT_RR_PATH_DEPTH"]
    assert define("RPTR_BINNED_LIGHTS_BIN_MAX_SIZE") == abi.BINNED_LIGHTS_BIN_MAX_SIZE == ref["BINNED_LIGHTS_BIN_MAX_SIZE"]
    for k in ("UNIFORM", "BN", "SOBOL", "Z_SBL"):
        assert define("RPTR_RNG_VARIANT_" + k) == getattr(abi, "RNG_VARIANT_" + k) == ref["RNG_VARIANT_" + k]
    for k in ("NOALPHA", "ONESIDED", "VOLUME", "EXTENDED"):
        assert define("RPTR_BASE_MATERIAL_" + k) == ref["BASE_MATERIAL_" + k]
    # the (2, 3) Halton table of the raster-TAA jitter (librender/halton.h compiled where it lies) against the oracle's regenerated one
    import oracle_lib as O
    for k in range(16):
        h = O.halton23(k)
        assert h[0] == np.float32(ref["halton_23_%d_x" % k]) and h[1] == np.float32(ref["halton_23_%d_y" % k]), k
    assert define("RPTR_SOBOL_TABLE_BYTES") == abi.SOBOL_TABLE_BYTES == ref["sizeof_SobolData"]
    assert ref["offsetof_SobolData_tile_invert_1_0"] == 1024 * 32 * 4
    assert define("RPTR_BN_TABLE_MIN_BYTES") == abi.BN_TABLE_MIN_BYTES == ref["offsetof_BNData_tile_scrambling_yx_d_4spp"]     # the 1 spp prefix of BNData
    assert [define("RPTR_AOV_" + k) for k in ("ALBEDO_ROUGHNESS", "NORMAL_DEPTH", "MOTION_JITTER")] == \
        [ref["OUTPUT_CHANNEL_" + k] - 1 for k in ("ALBEDO_ROUGHNESS", "NORMAL_DEPTH", "MOTION_JITTER")]
    assert (ref["REPROJECTION_MODE_NONE"], ref["REPROJECTION_MODE_DISCARD_HISTORY"]) == (0, 1)            # kernels_misc.h rp_k_resolve, oracle.cpp
    assert ref["default_RenderBackendOptions_rebuild_triangle_budget"] == 500000 and ref["default_RenderBackendOptions_force_bvh_rebuild"] == 0
    assert ref["DEFAULT_RAY_QUERY_BUDGET"] == 512 * 512
    # the dimension map of the table point sets (csrc/dshade.h RP_DIM_*, kernels.h draw sites; oracle/oshade.h DIM_*)
    dsh = open(os.path.join(os.path.dirname(build.LIB_PATH), "csrc", "dshade.h")).read()
    assert int(re.search(r"#define RP_DIM_CAMERA_END (\d+)u", dsh).group(1)) == ref["DIM_CAMERA_END"]
    assert int(re.search(r"#define RP_DIM_BOUNCE (\d+)u", dsh).group(1)) == ref["DIM_VERTEX_END"] + ref["DIM_LIGHT_END"]
    assert (ref["DIM_PIXEL_X"], ref["DIM_DIRECTION_X"], ref["DIM_LOBE"], ref["DIM_RR"], ref["DIM_LIGHT_SEL_1"], ref["DIM_POSITION_X"]) == (0, 0, 2, -1, 0, 2)
    # texture handles (rendering/bsdfs/texture_channel_mask.h)
    bits = struct.unpack("<I", struct.pack("<f", abi.textured_param(1234, 2)))[0]
    assert bits == ref["texture_handle_1234_2"] and (bits & 0x1FFFFFFF, (bits >> 29) & 3) == (ref["texture_handle_id"], ref["texture_handle_channel"])
    assert (ref["STANDARD_TEXTURE_NORMAL_SLOT"], ref["STANDARD_TEXTURE_BASECOLOR_SLOT"], ref["STANDARD_TEXTURE_SPECULAR_SLOT"]) == (0, 1, 2)


def _tool(*args):
    import subprocess
    return subprocess.run(["bash"] + list(args), capture_output=True, text=True, cwd=ROOT, timeout=300).stdout


def test_traversal_kernels_keep_their_register_budgets_and_their_node_fetch():
    """Facts of the built code object the design leans on and a compiler update could silently change (csrc/dtraverse.h):
    the benchmarked traversal instantiations -- one instance record, no alpha test -- fit the register budgets of seven (closest hit) and eight
    (shadow rays) waves per SIMD without scratch, hold 20 KB of LDS stacks per block, and fetch a node as 16 + 16 + 16 + 8 bytes (the
    vectoriser once made the last load 16: 8 bytes of padding per lane and node visit, profiles/r06_notes.md section 9)."""
    import shutil
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump") or shutil.which("bash") is None:
        pytest.skip("no llvm-objdump")
    if not os.path.exists(build.LIB_PATH):
        build.build_library()
    regs = {}
    for line in _tool("tools/kernel_regs.sh", build.LIB_PATH).splitlines():
        m = re.match(r"(\S+)\s+vgpr\s+(\d+)\s+sgpr\s+(\d+)\s+scratch\s+(\d+)\s+lds\s+(\d+)", line)
        if m:
            regs[m.group(1)] = tuple(int(x) for x in m.groups()[1:])
    budgets = {"_Z11rp_k_extendILb0ELb1ELb0ELb1ELb0EE": 72, "_Z11rp_k_extendILb0ELb0ELb0ELb1ELb0EE": 72, "_Z12rp_k_connectILb0ELb0ELb1EE": 64}
    for prefix, budget in budgets.items():
        hits = [v for k, v in regs.items() if k.startswith(prefix)]
        assert len(hits) == 1, (prefix, sorted(regs)[:5])
        vgpr, _, scratch, lds = hits[0]
        assert vgpr <= budget and scratch == 0 and lds == 20 * 256 * 4, (prefix, hits[0])
    isa = _tool("tools/disasm.sh", build.LIB_PATH, "_Z11rp_k_extendILb0ELb0ELb0ELb1ELb0EE")
    loads = re.findall(r"global_load_(dwordx?\d?)\s+v\[?[0-9:]+\]?, (v\d+), (s\[\d+:\d+\])(?: offset:(\d+))?", isa)
    # the node fetch: four loads off one 32-bit offset register and one scalar base, at byte 0 / 16 / 32 / 48 of the node
    by_base = {}
    for width, vaddr, sbase, off in loads:
        by_base.setdefault((vaddr, sbase), []).append((int(off or 0), width))
    node = [sorted(v) for v in by_base.values() if sorted(o for o, _ in v) == [0, 16, 32, 48]]
    assert node and all(n == [(0, "dwordx4"), (16, "dwordx4"), (32, "dwordx4"), (48, "dwordx2")] for n in node), by_base
