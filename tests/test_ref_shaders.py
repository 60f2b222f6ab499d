"""Golden vectors from the reference's OWN shader library (rendering/*.glsl compiled as C++ against GLM by oracle/ref_shader_driver.cpp:
`make -C oracle ref_shaders GLM_ROOT=<dir>` writes tests/golden/ref_shaders.json) against the oracle's restatement of the same functions.

This is the pin the oracle lacks for the functions that decide a pixel ("parity unpinned" for BSDFs and light
sampling). The build image has no GLM and the rules forbid stand-in headers, so the fixture cannot be produced here: until somebody runs
the one command on a machine that has GLM, every test in this file SKIPS with that reason. With the fixture present they need no GLM, no
reference checkout and no GPU (the GPU side is pinned through the oracle by the whole-frame parity tests).

Tolerance: the oracle fixes one evaluation order per GLSL built-in (oracle/ovec.h: dot = (xx' + yy') + zz', normalize = v * (1 / sqrt(dot)))
where GLM is free to choose another, so values are compared to 2e-5 relative (+ 1e-6 absolute), discrete decisions (which lobe, which
light) must agree except where an input lies within that tolerance of the decision boundary (at most 1 % of the vectors)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, scenes

FIXTURE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_shaders.json")
pytestmark = pytest.mark.skipif(not os.path.exists(FIXTURE), reason="tests/golden/ref_shaders.json absent: it is written by `make -C oracle ref_shaders "
                                "GLM_ROOT=<dir with glm/glm.hpp>` from the reference's own rendering/*.glsl; this image has no GLM")
RTOL, ATOL = 2e-5, 1e-6


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.fixture(scope="module")
def ref():
    return json.load(open(FIXTURE))


def _close(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.isclose(a, b, rtol=RTOL, atol=ATOL) | both_nan


def test_rng_against_the_references_generator(ref):
    state, first = O.rng_probe(3, 7, 10, 20, 256, n=1)   # (SURVEY 8a2's tuple: oracle/ref_shader_driver.cpp)
    assert int(state) == int(ref["rng"]["state"]) and np.float32(first[0]) == np.float32(ref["rng"]["first"])


def test_gltf_bsdf_sample_eval_pdf_against_the_references_functions(ref):
    L = O.lib()
    rows = ref["gltf"]
    bad = 0
    for r in rows:
        m = abi.BaseMaterial()
        m.base_color[:] = r["base_color"]
        m.normal_map = -1
        m.flags = abi.BASE_MATERIAL_NOALPHA
        m.metallic, m.specular, m.roughness, m.ior = r["metallic"], r["specular"], r["roughness"], r["ior"]
        n, wo = (np.array([r[k]], np.float32) for k in ("n", "wo"))
        u = np.array([r["u"]], np.float32)
        wi, w, f = (np.zeros((1, 3), np.float32) for _ in range(3))
        pdf, mis, wpdf = (np.zeros(1, np.float32) for _ in range(3))
        L.orc_gltf_sample(C.byref(m), _p(n), _p(wo), _p(u), 1, _p(wi), _p(w), _p(pdf), _p(mis), _p(f), _p(wpdf))
        ok = _close(wi[0], r["wi"]).all() and _close(w[0], r["weight"]).all() and _close(pdf[0], r["pdf"]) and _close(mis[0], r["mis_pdf"])
        wie = np.array([r["wi_eval"]], np.float32)
        fe, pe = np.zeros((1, 3), np.float32), np.zeros(1, np.float32)
        L.orc_gltf_eval(C.byref(m), _p(n), _p(wo), _p(wie), 1, _p(fe), _p(pe))
        assert _close(fe[0], r["f"]).all() and _close(pe[0], r["wpdf"]), r      # evaluation: no discrete decision, no excuse
        bad += 0 if ok else 1
    assert bad <= len(rows) // 100, "%d of %d sampled directions differ (lobe decisions at a boundary are the only excuse)" % (bad, len(rows))


def test_binned_ris_light_sampling_against_the_references_function(ref):
    L = O.lib()
    L.orc_sample_tri_lights.argtypes = None
    lights = np.array(ref["lights"], np.float32).reshape(-1, 4, 3)
    s = scenes.cornell32()
    s.lights = lights
    osc = O.OracleScene(s)
    bad = 0
    rows = ref["tri_lights"]
    for r in rows:
        cfg = abi.LightSamplingConfig(0.0, int(r["bin_size"]), 15.0, 0.0)
        p, n = (np.array([r[k]], np.float32) for k in ("p", "n"))
        u = np.array([r["u"]], np.float32)
        Lr, d = np.zeros((1, 3), np.float32), np.zeros((1, 3), np.float32)
        dist, pdf, mis = (np.zeros(1, np.float32) for _ in range(3))
        L.orc_sample_tri_lights(C.c_void_p(osc.h), C.byref(cfg), _p(p), _p(n), _p(u), 1, _p(Lr), _p(d), _p(dist), _p(pdf), _p(mis))
        ok = (_close(Lr[0], r["radiance_over_pdf"]).all() and _close(d[0], r["dir"]).all() and _close(dist[0], r["dist"]) and _close(pdf[0], r["pdf"])
              and _close(mis[0], r["mis_wpdf"]))
        bad += 0 if ok else 1
    assert bad <= len(rows) // 100, "%d of %d light samples differ" % (bad, len(rows))
