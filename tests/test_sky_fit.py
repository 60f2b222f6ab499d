"""Row a15, host half: host/sky_fit.hpp and sky_fit.py (restatements of vulkan/render_sky.cpp:25-72 around
rendering/lights/sky_model_arhosek/sky_model.cpp, reading the model's data headers at run time) against the reference's own code
compiled in place (oracle/_ref/libsky_ref.so) -- bit for bit on random scene states. The data only exist in the build container
(/root/reference), so these tests run there; the GPU-side test (tests/test_validation_cli.py) uses synthetic tables of the same layout."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from realtimepathtracingresearchframework_amd import abi, sky_fit

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_DATA = "/root/reference/rendering"
SKY_REF = os.path.join(ROOT, "oracle", "_ref", "libsky_ref.so")
needs_reference = pytest.mark.skipif(not (os.path.isdir(REF_DATA) and os.path.exists(SKY_REF)),
                                     reason="the Hosek-Wilkie data headers and oracle/_ref/libsky_ref.so exist in the build container only")

SHIM = r'''
#include "sky_fit.hpp"
extern "C" int shim_fit(const char *where, const float *sun_dir, float turbidity, const float *albedo, int light_count, RptrSceneParams *out, char *err, int cap) {
    static rptr::SkyTables t;
    static std::string loaded;
    std::string e;
    if (loaded != where) {
        t = rptr::SkyTables();
        if (!rptr::load_sky_tables(where, t, e)) { snprintf(err, cap, "%s", e.c_str()); return 1; }
        loaded = where;
    }
    snprintf(err, cap, "%s", e.c_str());
    return rptr::fit_sky(t, sun_dir, turbidity, albedo, light_count, *out) ? 0 : 2;
}
extern "C" void shim_sun(float h, float a, float *out) { rptr::sun_dir_from_height_angle(h, a, out); }
'''


class RefSkyOut(C.Structure):
    _fields_ = [("configs", (C.c_float * 4) * 9), ("radiances", C.c_float * 4), ("sun_dir", C.c_float * 3), ("sun_cos_angle", C.c_float),
                ("sun_radiance", C.c_float * 4)]


def build_shim(tmp_path):
    src = tmp_path / "sky_shim.cpp"
    src.write_text(SHIM)
    so = str(tmp_path / "libsky_shim.so")
    # (the reference side is built with plain -O2: no contraction on x86-64 without -mfma, same here)
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "host"),
                           str(src), "-o", so])
    L = C.CDLL(so)
    L.shim_fit.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float), C.c_int, C.POINTER(abi.SceneParams), C.c_char_p, C.c_int]
    return L


def states(n, seed=5):
    rng = np.random.default_rng(seed)
    out = [((0.0, 1.0, 0.0), 3.0, (0.2, 0.2, 0.2)), ((0.3, 0.8, 0.5), 10.0, (1.0, 1.0, 1.0)), ((1.0, 1e-3, 0.0), 1.0, (0.0, 0.0, 0.0)),
           ((0.2, -0.3, 0.9), 3.0, (0.2, 0.2, 0.2)), ((0.5, 0.5, 0.5), 9.999, (0.3, 0.6, 0.1)), ((0.1, 0.2, -0.7), 4.0, (0.5, 0.5, 0.5))]
    while len(out) < n:
        d = rng.normal(size=3)
        if rng.uniform() < 0.85:
            d[1] = abs(d[1])
        t = float(rng.integers(1, 11)) if rng.uniform() < 0.2 else float(rng.uniform(1.0, 10.0))
        out.append((tuple(float(v) for v in d), t, tuple(float(v) for v in rng.uniform(0, 1, 3))))
    return out


def as_words(sp):
    return np.frombuffer(bytes(sp), np.uint32)[:(160 + 12 + 4 + 16) // 4]   # sky params, sun_dir, sun_cos_angle, sun_radiance


def reference_params(R, d, t, al, lights):
    o = RefSkyOut()
    R.ref_update_sky_light((C.c_float * 3)(*d), C.c_float(t), (C.c_float * 3)(*al), lights, C.byref(o))
    sp = abi.SceneParams()
    for i in range(9):
        sp.sky_params.configs[i][:] = list(o.configs[i])
    sp.sky_params.radiances[:] = list(o.radiances)
    sp.sun_dir[:] = list(o.sun_dir)
    sp.sun_cos_angle = o.sun_cos_angle
    sp.sun_radiance[:] = list(o.sun_radiance)
    return sp


@needs_reference
def test_data_headers_are_parsed_completely():
    t = sky_fit.SkyTables(REF_DATA)
    assert [len(v) for v in t.rgb] == [1080] * 3 and [len(v) for v in t.rgb_rad] == [120] * 3
    assert t.has_sun and [len(v) for v in t.spec] == [1080] * 11 and [len(v) for v in t.solar] == [1800] * 11 and [len(v) for v in t.limb] == [6] * 11
    assert len(t.cie) == 3 * 95
    # the directory of the headers themselves works as well as the reference's rendering/ directory; without the spectral data the sun stays dark
    assert sky_fit.SkyTables(REF_DATA + "/lights/sky_model_arhosek").rgb == t.rgb


@needs_reference
def test_cpp_and_python_fit_equal_the_references_code_bit_for_bit(tmp_path):
    L = build_shim(tmp_path)
    R = C.CDLL(SKY_REF)
    tables = sky_fit.SkyTables(REF_DATA)
    err = C.create_string_buffer(512)
    n_lit = 0
    for k, (d, t, al) in enumerate(states(64)):
        lights = 3 if k % 3 == 0 else 0
        want = reference_params(R, d, t, al, lights)
        got = abi.SceneParams()
        rc = L.shim_fit(REF_DATA.encode(), (C.c_float * 3)(*d), C.c_float(t), (C.c_float * 3)(*al), lights, C.byref(got), err, 512)
        assert rc == 0, err.value
        assert np.array_equal(as_words(got), as_words(want)), (k, d, t, al)
        py = sky_fit.fit_sky(tables, d, t, al, lights)
        assert np.array_equal(as_words(py), as_words(want)), (k, d, t, al)
        n_lit += want.sun_radiance[0] > 0
    assert n_lit > 40    # most states have the sun above the horizon: its radiance went through the spectral model


@needs_reference
def test_fit_reproduces_the_packaged_sky_configurations():
    """data/sky_params.json (what the built-in scenes read) came from the reference's code: the run-time fit gives the same numbers"""
    from realtimepathtracingresearchframework_amd import scenes
    tables = sky_fit.SkyTables(REF_DATA)
    for key, cfg in scenes.SKY_CONFIGS.items():
        for lights in (False, True):
            fx = scenes.load_sky_fixture(key, has_lights=lights)
            sp = sky_fit.fit_sky(tables, cfg["sun_dir"], cfg["turbidity"], cfg["albedo"], 1 if lights else 0)
            got = np.array([list(r) for r in sp.sky_params.configs], np.float32)[:, :3]
            assert np.array_equal(got, np.array(fx["configs"], np.float32)[:, :3], equal_nan=True), key      # (the night sky is NaN in the reference too)
            assert np.array_equal(np.float32(sp.sky_params.radiances[:3]), np.float32(fx["radiances"][:3]), equal_nan=True)
            assert list(np.float32(sp.sun_radiance[:])) == list(np.float32(fx["sun_radiance"])) and list(np.float32(sp.sun_dir[:])) == list(np.float32(fx["sun_dir"]))


def test_sun_sliders_and_missing_data(tmp_path):
    L = build_shim(tmp_path)
    out = (C.c_float * 3)()
    for h, a in ((90.0, 0.0), (30.0, 45.0), (0.0, -180.0), (61.5, 170.25)):
        L.shim_sun(C.c_float(h), C.c_float(a), out)
        assert np.array_equal(np.array(list(out), np.float32), sky_fit.sun_dir_from_height_angle(h, a))
        assert abs(np.degrees(np.arcsin(out[1])) - h) < 1e-3
    err = C.create_string_buffer(512)
    sp = abi.SceneParams()
    assert L.shim_fit(str(tmp_path).encode(), (C.c_float * 3)(0, 1, 0), C.c_float(3.0), (C.c_float * 3)(0.2, 0.2, 0.2), 0, C.byref(sp), err, 512) == 1
    assert b"sky_model_data_rgb.h" in err.value
    with pytest.raises(FileNotFoundError):
        sky_fit.SkyTables(str(tmp_path))
