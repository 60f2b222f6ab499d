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
