"""Every BASELINE.json config at the LETTER of its configuration, the WHOLE frame against the oracle (the full-size
tests of rounds 1-2 compared bands of 12-16 rows of 1080; the oracle renders a whole C2 frame in well under a second on the GPU box).

  C1  Cornell box (32 triangles), 256x256, 1 spp, through `bin/rptr_hip --validation <prefix> --validation-spp 1 --pfm`
      (libapp/app_state.cpp:464-481: <prefix>_%04d.pfm of the float accumulation buffer)
  C2  procedural 1 M-triangle mesh, 1920x1080, 4 spp, diffuse-only BSDF
  C3  the same scene + 512 emissive triangles, glTF BSDF + binned-RIS NEE, 1920x1080, 8 spp: two-level AND as one flattened tree
      (what bench.py times)
  C4  10 M-triangle instanced forest, 1920x1080, 4 spp: the two-level tree against the oracle's own tree AND the flattened
      world-space tree against the oracle walking the exported tree
  C5  animated 1 M-triangle scene, 3840x2160, 2 spp, after the last of several per-frame refits

Every configuration is rendered by BOTH builds of the shading arithmetic (option "fast_math", csrc/dmath.h): 0 = IEEE division / square root
(the default: the oracle's operations statement by statement), 1 = the hardware's 1-ulp reciprocal / square root. Both against the same
oracle image and to the same assertions; the IEEE build additionally to RMSE < 1e-4 (it measures 1e-8 ... 6e-5).

Every test prints RMSE / max-abs / the number of pixels off by more than 1e-3 and asserts: RMSE < 1e-3 (north_star's tolerance) over
the whole frame, identical NaN masks, identical coverage (the alpha channel: 0 where the camera ray left the scene), and equal ray
counts up to the branch flips an ulp of libm causes (1e-3 relative)."""
import glob
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from common import RMSE_TOL, gpu_render, image_error
from realtimepathtracingresearchframework_amd import abi, backend, scenes

pytestmark = pytest.mark.gpu


MATH_BUILDS = [0, 1]   # option "fast_math"
IEEE_RMSE_TOL = 1e-4    # the IEEE build (same operations as the oracle up to libm's transcendental functions)


def math_tag(m):
    return "fast_math=%d (%s)" % (m, "hardware rcp / sqrt / rsq" if m else "IEEE")


def compare_whole_frame(tag, got, ref, st=None, ost=None, coverage=True, rmse_tol=RMSE_TOL):
    rmse, same_nan, maxabs = image_error(got, ref)
    d = np.abs(got[..., :3] - ref[..., :3])
    off = int((np.nan_to_num(d, nan=0.0).max(axis=2) > 1e-3).sum())
    line = "%s: %dx%d whole frame vs oracle: RMSE %.3g  max-abs %.3g  pixels off by > 1e-3: %d of %d" % (
        tag, got.shape[1], got.shape[0], rmse, maxabs, off, got.shape[0] * got.shape[1])
    if st is not None and ost is not None:
        line += "  | rays closest %d / %d, shadow %d / %d (GPU / oracle)" % (st.raw.rays_closest, ost.rays_closest, st.raw.rays_shadow, ost.rays_shadow)
    print(line)
    assert same_nan, tag + ": NaN masks differ"
    assert rmse < rmse_tol, line
    if coverage:
        assert np.array_equal(got[..., 3], ref[..., 3]), tag + ": coverage (alpha) differs in %d pixels" % int((got[..., 3] != ref[..., 3]).sum())
    if st is not None and ost is not None:
        assert abs(int(st.raw.rays_closest) - int(ost.rays_closest)) <= 1e-3 * ost.rays_closest
        assert abs(int(st.raw.rays_shadow) - int(ost.rays_shadow)) <= 1e-3 * max(ost.rays_shadow, 1)
    return rmse


# ---------------------------------------------------------------- C1
def test_c1_cornell_256x256_1spp_through_the_validation_cli(tmp_path):
    from test_validation_cli import _build_cli, read_pfm
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    assert s.num_tris() == 32
    path = str(tmp_path / "cornell.rpsc")
    s.dump(path)
    W = H = 256
    prefix = str(tmp_path / "c1")
    p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", "1", "--img", str(W), str(H), "--pfm"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    files = sorted(glob.glob(prefix + "_*.pfm"))
    assert [os.path.basename(f) for f in files] == ["c1_0001.pfm"], files      # validation naming: <prefix>_%04d of the accumulated spp
    img = read_pfm(files[0])
    assert img.shape == (H, W, 3)
    osc = O.OracleScene(s)
    ref, ost = osc.render(W, H, 1, variant=abi.VARIANT_GLTF)
    # the PFM holds RGB; coverage comes from the same backend through the Python mirror, which must hold the very same RGB bits
    got, st, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF)
    assert np.array_equal(img.view(np.uint32), np.ascontiguousarray(got[..., :3]).view(np.uint32))
    compare_whole_frame("C1 cornell-32", got, ref, st, ost, rmse_tol=IEEE_RMSE_TOL)
    # ... and the other build of the shading arithmetic, through the same program (the option's environment override)
    prefix = str(tmp_path / "c1f")
    p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", "1", "--img", str(W), str(H), "--pfm"], capture_output=True, text=True,
                       env=dict(os.environ, RPTR_FAST_MATH="1"))
    assert p.returncode == 0, p.stderr
    imgf = read_pfm(sorted(glob.glob(prefix + "_*.pfm"))[0])
    gotf, stf, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF, options={"fast_math": 1})
    assert np.array_equal(imgf.view(np.uint32), np.ascontiguousarray(gotf[..., :3]).view(np.uint32))
    compare_whole_frame("C1 cornell-32, " + math_tag(1), gotf, ref, stf, ost)


# ---------------------------------------------------------------- C2
def test_c2_whole_frame_1m_triangles_1080p_4spp_diffuse():
    s = scenes.grid_1m()
    assert s.num_tris() == 1_000_000
    W, H, spp = 1920, 1080, 4
    osc = O.OracleScene(s)
    osc.build_bvh()
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_SIMPLE)
    for m in MATH_BUILDS:
        got, st, _ = gpu_render(s, W, H, spp, abi.VARIANT_SIMPLE, options={"fast_math": m})
        compare_whole_frame("C2 grid-1M diffuse, " + math_tag(m), got, ref, st, ost, rmse_tol=RMSE_TOL if m else IEEE_RMSE_TOL)


# ---------------------------------------------------------------- C3
@pytest.mark.library_defaults
@pytest.mark.parametrize("flatten", [0, None])
def test_c3_whole_frame_gltf_area_lights_1080p_8spp(flatten, monkeypatch):
    """flatten=None: the library's default, which is what bench.py times -- the height field and the emitter mesh (two identity instances)
    in ONE tree instead of a top level over two bottom-level trees, without the host asking for it (option "flatten" = auto). An identity
    transform moves no vertex, so the oracle's own tree finds the same hits either way."""
    if flatten is not None:
        monkeypatch.setenv("RPTR_FLATTEN", str(flatten))
    s = scenes.grid_1m_lights()
    assert s.num_tris() == 1_000_512 and len(s.lights) >= 512 and len(s.instances) == 2
    W, H, spp = 1920, 1080, 8
    osc = O.OracleScene(s)
    osc.build_bvh()
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF)
    for m in MATH_BUILDS:
        got, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True, options={"fast_math": m} if m else None)   # (m = 0: nothing set at all)
        info = r.bvh_build_info()
        flat = bool(np.frombuffer(np.ascontiguousarray(r.export_bvh()[2]).tobytes(), np.int32).reshape(-1, 32)[0, 15] & 1)   # RPTR_BVH_INSTANCE_FLAT on record 0
        assert flat == (flatten is None) and r.get_option("flatten") == (-1 if flatten is None else 0) and r.get_option("fast_math") == m
        r.close()
        print("C3 flatten=%s: %s" % (flatten, info))
        compare_whole_frame("C3 grid-1M glTF + 512 emitters (%s), %s" % ("one flattened tree" if flatten is None else "two-level", math_tag(m)), got, ref, st, ost,
                            rmse_tol=RMSE_TOL if m else IEEE_RMSE_TOL)


# ---------------------------------------------------------------- C4
@pytest.mark.library_defaults
@pytest.mark.parametrize("flatten", [0, None])
def test_c4_whole_frame_forest_10m_instanced_triangles_1080p_4spp(flatten, monkeypatch):
    """flatten=None: the library's default -- a static forest is flattened (and built on the device) without the host asking for it"""
    if flatten is not None:
        monkeypatch.setenv("RPTR_FLATTEN", str(flatten))
    flatten = 1 if flatten is None else 0
    s = scenes.forest()
    assert s.num_instanced_tris() == 10_000_002 and len(s.instances) == 1001
    W, H, spp = 1920, 1080, 4
    got, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True)
    assert bool(np.frombuffer(np.ascontiguousarray(r.export_bvh()[2]).tobytes(), np.int32).reshape(-1, 32)[0, 15] & 1) == bool(flatten)
    osc = O.OracleScene(s)
    if flatten:
        # hits are found on world-space triangles: the oracle walks the very tree the device walked (t / u / v bit for bit)
        osc.import_bvh(*r.export_bvh())
        mode = O.BVH_IMPORTED
    else:
        osc.build_bvh()   # the oracle's own two-level binary tree: same hit per ray (closest t, ties by ids), whatever the tree
        mode = O.BVH_OWN
    r.close()
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, bvh_mode=mode)
    what = "flattened, oracle on the exported tree" if flatten else "two-level, oracle on its own tree"
    compare_whole_frame("C4 forest-10M %s, %s" % (what, math_tag(0)), got, ref, st, ost, rmse_tol=IEEE_RMSE_TOL)
    # (the device-built tree of the flattened forest is deterministic: the second handle walks the tree the oracle imported)
    gotf, stf, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, options={"fast_math": 1})
    compare_whole_frame("C4 forest-10M %s, %s" % (what, math_tag(1)), gotf, ref, stf, ost)


# ---------------------------------------------------------------- C5
def test_c5_whole_frame_animated_4k_2spp_after_the_last_refit():
    import torch
    NX, NZ = 1000, 500
    s = scenes.grid(NX, NZ, deform_t=0.0, name="grid-1M-dynamic")
    assert s.num_tris() == 1_000_000
    W, H, spp = 3840, 2160, 2
    times = [k / 60 for k in range(1, 5)]
    images = {}
    for m in MATH_BUILDS:
        r = backend.RenderHip(frames_in_flight=3, options={"fast_math": m})
        r.initialize(W, H)
        r.set_scene(s)
        cam = s.camera_params()
        queue, last, st = [], np.zeros((H, W, 4), np.float32), None
        for t in times:   # every frame: new vertices on the device, refit, render (frames in flight as bench.py --animate runs them)
            buf = torch.from_numpy(np.ascontiguousarray(scenes.grid_positions(NX, NZ, t), dtype=np.float32)).cuda()
            torch.cuda.synchronize()
            r.update_vertices_device(0, buf.data_ptr(), buf.shape[0])
            r.refit()
            queue.append(r.render_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True), spp=spp))
            torch.cuda.synchronize()   # (the buffer is borrowed until the copy has run)
            if len(queue) >= 3:
                st = r.wait(queue.pop(0))
        for ticket in queue:
            st = r.wait(ticket)
        assert r.readback_framebuffer(last) == W * H * 4
        r.close()
        images[m] = (last, st)
    osc = O.OracleScene(s)
    osc.set_dynamic_vertices(0, scenes.grid_positions(NX, NZ, times[-1]))
    osc.build_bvh()
    # the 4th reset of the handle: frame_offset = 3 frames x 2 samples (begin_frame's rule, render_vulkan.cpp:1937-1941)
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_SIMPLE, frame_offset=(len(times) - 1) * spp)
    for m in MATH_BUILDS:
        compare_whole_frame("C5 animated grid-1M, frame %d, %s" % (len(times), math_tag(m)), images[m][0], ref, images[m][1], ost, rmse_tol=RMSE_TOL if m else IEEE_RMSE_TOL)
