"""host/rptr_validate.cpp: the reference's headless `--validation <prefix> --validation-spp n --img w h --pfm` run
(cmdline.cpp:42-50, libapp/app_state.cpp:464-481, util/write_image.cpp:34-64) through the C ABI, on a scene dump."""
import os
import subprocess

import numpy as np
import pytest

from realtimepathtracingresearchframework_amd import abi, build, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "realtimepathtracingresearchframework_amd", "host")


def _build_cli(tmp_path):
    if not os.path.exists(build.LIB_PATH):
        build.build_library()
    exe = str(tmp_path / "rptr_validate")
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(HOST, "rptr_validate.cpp"), "-o", exe, "-L" + libdir,
                           "-lrptr_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def read_pfm(path):
    """the reference's layout: 'PF', 'w h', '-1.0' (little endian), rows bottom-up, RGB float32"""
    with open(path, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = (int(v) for v in f.readline().split())
        assert f.readline() == b"-1.0\n"
        data = np.frombuffer(f.read(), dtype="<f4")
    assert data.size == w * h * 3
    return data.reshape(h, w, 3)[::-1]


def test_scene_dump_round_trips_through_the_cpp_loader(tmp_path):
    exe = _build_cli(tmp_path)
    s = scenes.grid(20, 10, with_emitters=True)
    path = str(tmp_path / "grid.rpsc")
    s.dump(path)
    out = subprocess.run([exe, path, "--describe"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    kv = dict(zip(out.stdout.split()[0::2], out.stdout.split()[1::2]))
    assert int(kv["geometries"]) == len(s.geometries) and int(kv["meshes"]) == len(s.meshes)
    assert int(kv["parameterized_meshes"]) == len(s.pmeshes) and int(kv["instances"]) == len(s.instances)
    assert int(kv["materials"]) == len(s.materials) and int(kv["lights"]) == len(s.lights) == 512
    assert int(kv["triangles"]) == s.num_tris()
    assert int(kv["qsum"]) == sum(int((np.asarray(g.qpos, np.uint64) & np.uint64(0xFFFFFF)).sum()) for g in s.geometries)
    assert abs(float(kv["fovy"]) - s.camera_params().fovy) < 1e-5
    assert int(kv["max_path_depth"]) == abi.RenderParams.default().max_path_depth
    assert int(kv["bin_size"]) == abi.LightSamplingConfig.default().bin_size
    # a truncated file is an error, not garbage
    with open(path, "rb") as f:
        blob = f.read()
    with open(path, "wb") as f:
        f.write(blob[:len(blob) // 2])
    assert subprocess.run([exe, path, "--describe"], capture_output=True).returncode == 3


def test_validation_cli_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    exe = _build_cli(tmp_path)
    path = str(tmp_path / "c.rpsc")
    scenes.cornell32().dump(path)
    p = subprocess.run([exe, path, "--validation", str(tmp_path / "out"), "--validation-spp", "1", "--img", "32", "32", "--pfm"],
                       capture_output=True, text=True)
    assert p.returncode == 3 and "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_validation_pfm_matches_backend_and_oracle(tmp_path):
    """4 frames of 1 spp accumulated by the CLI = one 4 spp render through the Python mirror (bit-exact, same backend),
    and within the north_star tolerance of the oracle."""
    import oracle_lib as O
    from common import RMSE_TOL, gpu_render, image_error
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "cornell.rpsc")
    s.dump(path)
    W, H, spp = 96, 64, 4
    prefix = str(tmp_path / "val")
    p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--every-frame"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    for k in range(1, spp + 1):
        assert os.path.exists("%s_%04d.pfm" % (prefix, k))
    img = read_pfm("%s_%04d.pfm" % (prefix, spp))
    ref_gpu, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    assert np.array_equal(img.view(np.uint32), np.ascontiguousarray(ref_gpu[..., :3]).view(np.uint32))
    osc = O.OracleScene(s)
    ref, _ = osc.render(W, H, spp, variant=abi.VARIANT_GLTF)
    rgba = np.concatenate([img, np.ones((H, W, 1), np.float32)], axis=2)
    rmse, same, _ = image_error(rgba, ref)
    assert same and rmse < RMSE_TOL
