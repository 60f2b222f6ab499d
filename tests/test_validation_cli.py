"""host/rptr_cli.cpp (bin/rptr_hip): the reference's headless run modes through the C ABI, on a scene dump:
`--validation <prefix> --validation-spp n --img w h --pfm` and `--profiling <csv prefix> --profiling-fps f --profiling-img p`
(cmdline.cpp:42-104, libapp/app_state.cpp:464-498, libapp/benchmark_info.cpp:69-124, util/write_image.cpp:34-64)."""
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
    exe = str(tmp_path / "rptr_hip")
    libdir = os.path.dirname(build.LIB_PATH)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(HOST, "rptr_cli.cpp"), "-o", exe, "-L" + libdir,
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


def read_exr(path):
    """minimal reader of what write_image.hpp writes: single-part scan-line OpenEXR, no compression, channels A B G R of one
    type -> (h, w, 4) RGBA array (float32, or uint16 holding halfs)"""
    import struct
    raw = open(path, "rb").read()
    assert struct.unpack_from("<II", raw, 0) == (20000630, 2)
    at, attrs = 8, {}
    while raw[at] != 0:
        name_end = raw.index(b"\0", at)
        type_end = raw.index(b"\0", name_end + 1)
        size = struct.unpack_from("<I", raw, type_end + 1)[0]
        attrs[raw[at:name_end].decode()] = (raw[name_end + 1:type_end].decode(), raw[type_end + 5:type_end + 5 + size])
        at = type_end + 5 + size
    at += 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, names, types = attrs["channels"][1], [], []
    k = 0
    while ch[k] != 0:
        e = ch.index(b"\0", k)
        names.append(ch[k:e].decode())
        types.append(struct.unpack_from("<I", ch, e + 1)[0])
        k = e + 1 + 16
    assert names == ["A", "B", "G", "R"] and len(set(types)) == 1
    dt = np.float32 if types[0] == 2 else np.uint16
    offsets = struct.unpack_from("<%dQ" % h, raw, at)
    img = np.zeros((h, w, 4), dt)
    for y in range(h):
        yy, size = struct.unpack_from("<iI", raw, offsets[y])
        assert yy == y and size == w * 4 * np.dtype(dt).itemsize
        rows = np.frombuffer(raw, dtype=dt, count=4 * w, offset=offsets[y] + 8).reshape(4, w)
        img[y, :, 3], img[y, :, 2], img[y, :, 1], img[y, :, 0] = rows[0], rows[1], rows[2], rows[3]
    return img


def read_png(path):
    """8-bit RGBA PNG with filter type 0 rows (what write_image.hpp writes) -> (h, w, 4) uint8; checks every chunk's CRC"""
    import struct
    import zlib
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, w, h = 8, b"", 0, 0
    while at < len(raw):
        n, kind = struct.unpack_from(">I4s", raw, at)
        data = raw[at + 8:at + 8 + n]
        assert zlib.crc32(kind + data) == struct.unpack_from(">I", raw, at + 8 + n)[0]
        if kind == b"IHDR":
            w, h, depth, colour = struct.unpack_from(">IIBB", data, 0)
            assert (depth, colour) == (8, 6)
        elif kind == b"IDAT":
            idat += data
        at += 12 + n
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + 4 * w)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:].reshape(h, w, 4).copy()


def test_image_writers(tmp_path):
    """write_image.hpp on the host alone: PFM in the reference's layout, EXR (float and half) and PNG readable by the minimal
    readers above (PNG through zlib: stored deflate blocks, Adler-32 and CRCs are checked by it)"""
    src = tmp_path / "w.cpp"
    src.write_text('''
#include "write_image.hpp"
#include <cmath>
int main(int, char **argv) {
    const unsigned w = 37, h = 301;   // more than 65535 bytes of PNG rows: several stored blocks
    std::vector<float> f((size_t)w * h * 4);
    std::vector<uint16_t> half(f.size());
    std::vector<unsigned char> u8(f.size());
    for (size_t i = 0; i < f.size(); ++i) { f[i] = std::sin(0.37f * i) * 100.0f; half[i] = (uint16_t)(i * 7919u); u8[i] = (unsigned char)(i * 31u + 5u); }
    std::string p = argv[1];
    return rptr::write_pfm(p + "/a", w, h, 4, f.data()) && rptr::write_exr<float>(p + "/a", w, h, 4, f.data()) &&
           rptr::write_exr<uint16_t>(p + "/b", w, h, 4, half.data()) && rptr::write_png(p + "/a", w, h, 4, u8.data()) ? 0 : 1;
}
''')
    exe = str(tmp_path / "w")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + HOST, str(src), "-o", exe])
    assert subprocess.run([exe, str(tmp_path)]).returncode == 0
    w, h = 37, 301
    i = np.arange(w * h * 4, dtype=np.float32)
    f = (np.sin(np.float32(0.37) * i).astype(np.float32) * np.float32(100.0)).reshape(h, w, 4)
    exr = read_exr(str(tmp_path / "a.exr"))
    assert exr.dtype == np.float32 and np.allclose(exr, f, rtol=1e-5, atol=1e-4)      # (libm's sinf vs numpy's: not bit-equal)
    assert np.array_equal(read_pfm(str(tmp_path / "a.pfm")), exr[..., :3])            # the same floats in both files
    half = read_exr(str(tmp_path / "b.exr"))
    assert half.dtype == np.uint16 and np.array_equal(half.reshape(-1), (np.arange(w * h * 4, dtype=np.uint64) * 7919 % 65536).astype(np.uint16))
    png = read_png(str(tmp_path / "a.png"))
    assert np.array_equal(png.reshape(-1), ((np.arange(w * h * 4, dtype=np.uint64) * 31 + 5) % 256).astype(np.uint8))


def test_vks_scene_reaches_the_cpp_host_tool(tmp_path):
    """.vks -> `python -m ...vks --dump` -> bin/rptr_hip: the reference's asset format in front of the C++ host"""
    from realtimepathtracingresearchframework_amd import vks
    exe = _build_cli(tmp_path)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vks", "alpha_v4.vks")
    dump = str(tmp_path / "alpha.rpsc")
    assert vks.main([src, "--dump", dump, "--eye", "0.4", "1.3", "4.6", "--center", "0", "0.9", "0", "--fov", "45", "--sky", "low_sun"]) == 0
    s = vks.read_vks(src)
    out = subprocess.run([exe, dump, "--describe"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    kv = dict(zip(out.stdout.split()[0::2], out.stdout.split()[1::2]))
    assert int(kv["geometries"]) == len(s.geometries) == 7 and int(kv["instances"]) == 7 and int(kv["materials"]) == 6
    assert int(kv["triangles"]) == s.num_tris() and int(kv["lights"]) == len(s.lights)
    assert abs(float(kv["fovy"]) - 45.0) < 1e-5


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


@pytest.mark.gpu
def test_validation_accumulates_in_launch_sequences_with_the_bits_of_synchronous_frames(tmp_path):
    """--validation queues its frames as launch sequences of up to 16 samples, two sequences in flight (reset_rest = 0: a
    frame continues the accumulation of the one before it), instead of one synchronous frame per batch_spp samples: every image it writes
    -- with --every-frame one per frame, at the same sample counts -- has the bits the synchronous loop writes (--synchronous), for a
    target that is not a multiple of the sequence length and for frames of several samples."""
    exe = _build_cli(tmp_path)
    s = scenes.grid(60, 30, with_emitters=True)
    path = str(tmp_path / "grid.rpsc")
    s.dump(path)
    W, H = 120, 72
    for batch_spp, target in ((1, 37), (3, 30)):
        outs = {}
        # ("short": --frames-per-launch 2 --frames-in-flight 6, launch sequences of two frames with six in flight)
        for mode, extra in (("queued", []), ("sync", ["--synchronous"]), ("short", ["--frames-per-launch", "2", "--frames-in-flight", "6"])):
            prefix = str(tmp_path / ("val_%s_%d" % (mode, batch_spp)))
            p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", str(target), "--batch-spp", str(batch_spp), "--img", str(W), str(H), "--pfm",
                                "--every-frame"] + extra, capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            outs[mode] = prefix
        counts = list(range(batch_spp, target + batch_spp, batch_spp))
        assert counts[-1] >= target
        for k in counts:
            a, b = read_pfm("%s_%04d.pfm" % (outs["queued"], k)), read_pfm("%s_%04d.pfm" % (outs["sync"], k))
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (batch_spp, k)
            c = read_pfm("%s_%04d.pfm" % (outs["short"], k))
            assert np.array_equal(c.view(np.uint32), b.view(np.uint32)), ("frames per launch 2", batch_spp, k)


@pytest.mark.gpu
def test_queued_validation_fits_its_launch_sequences_to_the_sample_slots_of_a_large_frame(tmp_path):
    """above ~2.9 Mpixel per rank a frame context holds fewer than 16 sample slots (rptr_hip_get_option "sample_slots": 5 at 4K with
    the default path budget), and the queued --validation run used to ask for sequences of 16 samples regardless -- RPTR_E_INVALID, no image.
    The CLI now clamps a sequence to the slots: a 4K run with 2-sample frames completes, and writes the synchronous loop's bits."""
    exe = _build_cli(tmp_path)
    path = str(tmp_path / "c.rpsc")
    scenes.cornell32().dump(path)
    W, H = 3840, 2160
    outs = {}
    for mode, extra in (("queued", []), ("sync", ["--synchronous"])):
        prefix = str(tmp_path / ("big_" + mode))
        p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", "8", "--batch-spp", "2", "--img", str(W), str(H), "--pfm"] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr[-600:]
        outs[mode] = read_pfm(prefix + "_0008.pfm")
    assert outs["queued"].shape == (H, W, 3) and np.array_equal(outs["queued"].view(np.uint32), outs["sync"].view(np.uint32))
    from realtimepathtracingresearchframework_amd import backend
    r = backend.RenderHip()
    assert r.get_option("sample_slots") == 0            # (sized by initialize)
    r.initialize(W, H)
    assert 1 <= r.get_option("sample_slots") < 16
    r.close()


def test_cli_rejects_bad_mode_combinations(tmp_path):
    exe = _build_cli(tmp_path)
    path = str(tmp_path / "c.rpsc")
    scenes.cornell32().dump(path)
    # validation and profiling are mutually exclusive (cmdline.cpp:479-486); one of them is required
    assert subprocess.run([exe, path, "--validation", "a", "--profiling", "b"], capture_output=True).returncode == 2
    assert subprocess.run([exe, path], capture_output=True).returncode == 2
    assert subprocess.run([exe, path, "--validation"], capture_output=True).returncode == 2
    assert subprocess.run([exe, path, "--bogus"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_textured_scene_through_the_dump_and_the_cli(tmp_path):
    """textures travel in the scene dump: the C++ host renders the textured scene like the Python mirror, bit for bit"""
    from common import gpu_render
    exe = _build_cli(tmp_path)
    s = scenes.textured_test()
    path = str(tmp_path / "tex.rpsc")
    s.dump(path)
    prefix = str(tmp_path / "tex")
    p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", "2", "--img", "96", "72", "--pfm"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    img = read_pfm(prefix + "_0002.pfm")
    ref, _, _ = gpu_render(s, 96, 72, 2, abi.VARIANT_GLTF)
    assert np.array_equal(img.view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32))


@pytest.mark.gpu
def test_profiling_mode_csv_images_and_camera_flags(tmp_path):
    """profiling mode on a static scene: one CSV row per frame with the reference's header, frames accumulate, an image per
    second of animation time; --eye/--center/--fov move the camera (compared with the Python mirror)."""
    from common import gpu_render
    from realtimepathtracingresearchframework_amd import backend
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "cornell.rpsc")
    s.dump(path)
    W, H = 64, 48
    csv_prefix, img_prefix = str(tmp_path / "prof"), str(tmp_path / "pimg")
    p = subprocess.run([exe, path, "--profiling", csv_prefix, "--profiling-fps", "4", "--profiling-count", "8", "--profiling-img", img_prefix,
                        "--img", str(W), str(H), "--eye", "0.5", "0.2", "3.0", "--center", "0", "0", "0", "--fov", "50", "--pfm"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    rows = open(csv_prefix + ".csv").read().strip().split("\n")
    assert rows[0] == "frames_total,keyframe,frames_accumulated,render_time_ms,app_time_ms"
    vals = [r.split(",") for r in rows[1:]]
    assert [int(v[0]) for v in vals] == list(range(1, 9))
    assert [int(v[1]) for v in vals] == [1, 1, 1, 1, 2, 2, 2, 2]          # a keyframe = one second = 4 frames
    assert [int(v[2]) for v in vals] == list(range(1, 9))                   # static scene: the frames accumulate
    # the reference's loop: frames go through a command stream, two in flight, and a row's render_time_ms is what the backend's stats() holds
    # then -- the timings of the frame two submissions earlier (RenderStats::frame_stats_delay, vulkan/render_vulkan.cpp:2229-2243): none yet
    # for the first two frames (a read-back -- the image after frame 4 -- finishes what is in flight: frame 5 already sees frame 4's)
    assert [float(v[3]) > 0 for v in vals] == [False, False, True, True, True, True, True, True]
    assert "through a command stream (two frames in flight)" in p.stdout
    assert os.path.exists(img_prefix + "_0001.pfm") and os.path.exists(img_prefix + "_0002.pfm")
    # --synchronous (the application's "force synchronous rendering": cmd_stream = nullptr): every row has its own frame's time, same images
    p2 = subprocess.run([exe, path, "--profiling", csv_prefix + "_s", "--profiling-fps", "4", "--profiling-count", "8", "--profiling-img", img_prefix + "_s",
                         "--img", str(W), str(H), "--eye", "0.5", "0.2", "3.0", "--center", "0", "0", "0", "--fov", "50", "--pfm", "--synchronous"],
                        capture_output=True, text=True)
    assert p2.returncode == 0, p2.stderr
    assert all(float(r.split(",")[3]) > 0 for r in open(csv_prefix + "_s.csv").read().strip().split("\n")[1:]) and "synchronous" in p2.stdout
    for k in (1, 2):
        assert np.array_equal(read_pfm(img_prefix + "_%04d.pfm" % k).view(np.uint32), read_pfm(img_prefix + "_s_%04d.pfm" % k).view(np.uint32))
    # the last image holds 8 accumulated samples from the moved camera
    img = read_pfm(img_prefix + "_0002.pfm")
    cam = s.camera_params()
    eye, center = np.array([0.5, 0.2, 3.0], np.float32), np.zeros(3, np.float32)
    d = center - eye
    d = (d / np.float32(np.sqrt((d * d).sum(dtype=np.float32)))).astype(np.float32)
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    cam.pos[:] = [float(x) for x in eye]
    cam.dir[:] = [float(x) for x in d]
    cam.fovy = 50.0
    cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    r.render(cfg, spp=8)
    ref = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(ref)
    r.close()
    assert np.abs(img - ref[..., :3]).max() < 1e-4 * max(1.0, float(np.abs(ref[..., :3]).max()))


@pytest.mark.gpu
def test_profiling_mode_animated_wave_refits_every_frame(tmp_path):
    """SURVEY 8d C5 in the C++ host: --animate-wave moves the dynamic mesh before every frame, refit, accumulation restarts."""
    exe = _build_cli(tmp_path)
    s = scenes.grid(64, 32, deform_t=0.0)
    path = str(tmp_path / "grid.rpsc")
    s.dump(path)
    csv_prefix, img_prefix = str(tmp_path / "anim"), str(tmp_path / "aimg")
    p = subprocess.run([exe, path, "--profiling", csv_prefix, "--profiling-fps", "2", "--profiling-count", "4", "--profiling-img", img_prefix,
                        "--img", "96", "64", "--variant", "diffuse", "--animate-wave", "0.5", "0.4", "--pfm"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    vals = [r.split(",") for r in open(csv_prefix + ".csv").read().strip().split("\n")[1:]]
    assert [int(v[2]) for v in vals] == [1, 1, 1, 1]                        # every frame starts a new accumulation
    a, b = read_pfm(img_prefix + "_0001.pfm"), read_pfm(img_prefix + "_0002.pfm")
    assert np.isfinite(a).all() and np.isfinite(b).all() and not np.array_equal(a, b)   # the surface moved
    # a static mesh cannot be animated
    s2 = scenes.grid(16, 8)
    path2 = str(tmp_path / "static.rpsc")
    s2.dump(path2)
    p = subprocess.run([exe, path2, "--profiling", csv_prefix, "--animate-wave", "0.5", "0.4"], capture_output=True, text=True)
    assert p.returncode == 3 and "dynamic" in p.stderr


@pytest.mark.gpu
def test_exr_png_and_data_capture_outputs(tmp_path):
    """--exr (the default, as in the reference) holds the floats of --pfm plus alpha; --png the 8-bit frame buffer; --data-capture
    stores the accumulation buffer and the three AOV images as EXR (libapp/app_state.cpp:499-531) = what the Python mirror reads back"""
    from common import gpu_render
    from realtimepathtracingresearchframework_amd import backend
    exe = _build_cli(tmp_path)
    s = scenes.textured_test()
    path = str(tmp_path / "t.rpsc")
    s.dump(path)
    W, H, spp = 80, 60, 2
    common = ["--validation-spp", str(spp), "--img", str(W), str(H)]
    for flags in ([], ["--pfm"], ["--png"]):
        prefix = str(tmp_path / ("v" + "".join(flags).strip("-")))
        p = subprocess.run([exe, path, "--validation", prefix] + common + flags, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
    exr = read_exr(str(tmp_path / "v_0002.exr"))
    pfm = read_pfm(str(tmp_path / "vpfm_0002.pfm"))
    png = read_png(str(tmp_path / "vpng_0002.png"))
    assert exr.dtype == np.float32 and np.array_equal(exr[..., :3].view(np.uint32), pfm.view(np.uint32))
    img, _, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True)
    assert np.array_equal(exr.view(np.uint32), img.view(np.uint32))
    u8 = np.zeros((H, W, 4), np.uint8)
    r.readback_framebuffer(u8)
    assert np.array_equal(png, u8)
    # data capture: 1 spp, keyframe 1
    prefix = str(tmp_path / "cap")
    p = subprocess.run([exe, path, "--data-capture", prefix, "--data-capture-spp", "1", "--img", str(W), str(H)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    r2 = backend.RenderHip()
    r2.initialize(W, H)
    r2.set_scene(s)
    r2.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1)
    rgba = np.zeros((H, W, 4), np.float32)
    r2.readback_framebuffer(rgba)
    assert np.array_equal(read_exr(prefix + "_0001_rgba.exr").view(np.uint32), rgba.view(np.uint32))
    for k, name in enumerate(("albedo_roughness", "normal_depth", "motion_jitter")):
        half = np.zeros((H, W, 4), np.float16)
        r2.readback_aov(k, half)
        got = read_exr("%s_0001_%s.exr" % (prefix, name))
        assert got.dtype == np.uint16 and np.array_equal(got, half.view(np.uint16)), name
    r.close()
    r2.close()
    assert subprocess.run([exe, path, "--data-capture", prefix, "--validation", "x"], capture_output=True).returncode == 2



# ---------------------------------------------------------------- configuration / keyframe files, devices
INI_BASE = """[Window][Debug##Default]
Pos=60,60

[Application][]
target spp= 8
batch spp= 2
max path depth= 5
rr path depth= 3
glossy-only mode= 0
force bvh rebuild= 1
rebuild triangle budget= 250000
pixel radius= 1.000000e+00
[.][*output channel]
OUTPUT_CHANNEL_COLOR= 1
..
[.][*variant]
wavefront-gltf-transmission= 1
..
[.][*pointset]
Z_SBL= 1
..
[.][Filtering]
[.][*reprojection]
DISCARD_HISTORY= 1
..
use 2x upscaling= 0
raster TAA pattern= 1
unjittered raster pattern= 0
..

[Application][scene.vks]
[.][Camera]
speed= 1.000000e-01
position= 1.500000e+00 2.500000e-01 3.000000e+00
direction= -4.000000e-01 0.000000e+00 -9.165151e-01
up= 0.000000e+00 1.000000e+00 0.000000e+00
..
[.][Sensor]
light bin size= 8
..
[.][Tonemapping]
[.][*operator]
FAST_TONE_MAPPING= 1
..
exposure= 1.250000e+00
..
[.][Scene]
bump scale= 2.000000e+00
..
"""

INI_KEY = """[Application][scene.vks]
[.][Camera]
position= -1.000000e+00 0.000000e+00 3.200000e+00
..
[.][Tonemapping]
exposure= -5.000000e-01
..
"""


def test_cli_reads_the_reference_ini_configuration_files(tmp_path):
    """--config / --keyframe: the ImGui-settings text the reference serialises its state to (imstate.cpp:226-330): objects, nested
    headers, combo boxes, %e floats. --describe prints what the files make of the scene's defaults (no GPU needed)."""
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    (tmp_path / "base.ini").write_text(INI_BASE)
    (tmp_path / "key.ini").write_text(INI_KEY)
    out = subprocess.run([exe, path, "--describe", "--config", str(tmp_path / "base.ini"), "--keyframe", "0.5:" + str(tmp_path / "key.ini"),
                          "--keyframe", str(tmp_path / "base.ini")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    cfg = lines[1].split()
    kv = dict(zip(cfg[1::2], cfg[2::2]))
    assert (int(kv["target_spp"]), int(kv["batch_spp"]), int(kv["max_path_depth"]), int(kv["rr_path_depth"])) == (8, 2, 5, 3)
    assert abs(float(kv["exposure"]) - 1.25) < 1e-6 and int(kv["tonemap"]) == 2 and int(kv["output_channel"]) == 0
    assert int(kv["bin_size"]) == 8 and int(kv["variant"]) == abi.VARIANT_GLTF_TRANSMISSION
    assert int(kv["rng_variant"]) == abi.RNG_VARIANT_Z_SBL
    assert (int(kv["reprojection_mode"]), int(kv["raster_taa"]), int(kv["upscale"])) == (1, 1, 1)
    assert int(kv["force_bvh_rebuild"]) == 1 and int(kv["rebuild_triangle_budget"]) == 250000 and abs(float(kv["bump_scale"]) - 2.0) < 1e-6
    at = cfg.index("cam_pos")
    assert [float(v) for v in cfg[at + 1:at + 4]] == [1.5, 0.25, 3.0]
    k0, k1 = lines[2].split(), lines[3].split()
    assert float(k0[2]) == 0.5 and int(k0[4]) == 1 and abs(float(k0[6]) + 0.5) < 1e-6 and [float(v) for v in k0[8:11]] == [-1.0, 0.0, 3.2]
    assert float(k1[2]) == 1.0 and int(k1[4]) == 2
    # a missing file is the reference's error (main.cpp:131-133)
    bad = subprocess.run([exe, path, "--describe", "--config", str(tmp_path / "nope.ini")], capture_output=True, text=True)
    assert bad.returncode == 3 and "Cannot find config file" in bad.stderr
    assert subprocess.run([exe, path, "--validation", "x", "--backend", "vulkan"], capture_output=True).returncode == 2


@pytest.mark.gpu
def test_cli_keyframes_and_two_ranks_on_one_device(tmp_path):
    """profiling mode over two keyframes (0.5 s + 1 s at 4 fps = 6 frames, the accumulation restarts with the second keyframe, one
    image per keyframe), --freeze-frame repeats the frame, and --devices 0,0 (two ranks of the frame on one GPU, tiles gathered by the
    library) writes the image of a single device, bit for bit"""
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    (tmp_path / "base.ini").write_text(INI_BASE.replace("wavefront-gltf-transmission", "wavefront-gltf"))
    (tmp_path / "key.ini").write_text(INI_KEY)
    common = [exe, path, "--img", "96", "64", "--pfm"]
    run = subprocess.run(common + ["--profiling", str(tmp_path / "prof"), "--profiling-fps", "4", "--profiling-img", str(tmp_path / "pimg"),
                                   "--keyframe", "0.5:" + str(tmp_path / "base.ini"), "--keyframe", str(tmp_path / "key.ini")], capture_output=True, text=True)
    assert run.returncode == 0, run.stderr
    rows = (tmp_path / "prof.csv").read_text().strip().splitlines()
    assert rows[0] == "frames_total,keyframe,frames_accumulated,render_time_ms,app_time_ms" and len(rows) == 1 + 6
    cols = [r.split(",") for r in rows[1:]]
    assert [int(c[1]) for c in cols] == [1, 1, 2, 2, 2, 2] and [int(c[2]) for c in cols] == [1, 2, 1, 2, 3, 4]
    k1, k2 = read_pfm(str(tmp_path / "pimg_0001.pfm")), read_pfm(str(tmp_path / "pimg_0002.pfm"))
    assert k1.shape == (64, 96, 3) and not np.array_equal(k1, k2)          # another camera, another exposure-independent radiance
    # one device against two ranks on that device, validation mode, same configuration
    for tag, extra in (("one", []), ("two", ["--devices", "0,0", "--stripe-rows", "8"])):
        r = subprocess.run(common + ["--validation", str(tmp_path / tag), "--validation-spp", "4", "--config", str(tmp_path / "base.ini")] + extra,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    a, b = read_pfm(str(tmp_path / "one_0004.pfm")), read_pfm(str(tmp_path / "two_0004.pfm"))
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and a.std() > 0.01
    # --freeze-frame: every frame repeats the same samples -> accumulating the same frame changes nothing
    for tag, extra in (("f1", ["--validation-spp", "1"]), ("f3", ["--validation-spp", "3"])):
        r = subprocess.run(common + ["--validation", str(tmp_path / tag), "--freeze-frame"] + extra, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    f1, f3 = read_pfm(str(tmp_path / "f1_0001.pfm")), read_pfm(str(tmp_path / "f3_0003.pfm"))
    assert np.array_equal(f1.view(np.uint32), f3.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name,rng_variant", [("sobol", abi.RNG_VARIANT_SOBOL), ("z-sobol", abi.RNG_VARIANT_Z_SBL)])
def test_cli_point_sets_match_the_python_host(tmp_path, name, rng_variant):
    """--rng-variant: the C++ host derives the same SobolData from the package's matrices as pointsets.py (tile inversion included)
    -> the validation image of the CLI equals the Python host's, bit for bit; the blue-noise stand-in table renders too"""
    from realtimepathtracingresearchframework_amd import backend
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    W, H = 272, 64
    r = subprocess.run([exe, path, "--img", str(W), str(H), "--pfm", "--validation", str(tmp_path / "v"), "--validation-spp", "2", "--rng-variant", name],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_pfm(str(tmp_path / "v_0002.pfm"))
    rh = backend.RenderHip()
    rh.initialize(W, H)
    rh.set_scene(s)
    rh.set_rng_variant(rng_variant)
    for k in range(2):
        rh.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=(k == 0)), spp=1)
    img = np.zeros((H, W, 4), np.float32)
    rh.readback_framebuffer(img)
    rh.close()
    assert np.array_equal(got.view(np.uint32), img[..., :3].view(np.uint32))
    if rng_variant == abi.RNG_VARIANT_SOBOL:
        b = subprocess.run([exe, path, "--img", str(W), str(H), "--pfm", "--validation", str(tmp_path / "b"), "--validation-spp", "2", "--rng-variant", "bn"],
                           capture_output=True, text=True)
        assert b.returncode == 0, b.stderr
        bn = read_pfm(str(tmp_path / "b_0002.pfm"))
        assert np.isfinite(bn).all() and bn.std() > 0.01 and not np.array_equal(bn, got)
        assert subprocess.run([exe, path, "--validation", "x", "--rng-variant", "halton"], capture_output=True).returncode == 2


# ---------------------------------------------------------------- .vks read by the C++ host itself (a20, f2)
def _py_dump(vks_path, out):
    from realtimepathtracingresearchframework_amd import vks
    s = vks.read_vks(vks_path)
    s.dump(out)
    return s


@pytest.mark.parametrize("name", ["alpha_v4.vks", "alpha_v3.vks"])
def test_cpp_vks_reader_equals_the_python_reader(tmp_path, name):
    """host/vks_reader.hpp + host/lights.hpp + host/sky_params.hpp against vks.py + lights.py + scenes.py on the golden .vks files (which
    libvkr itself reads the same way, tests/test_vks.py): geometry streams, per-triangle material ids, instance transforms (version 3:
    quantised on reading), materials with texture handles, decoded BC textures, the binned emitters, default camera and sky --
    the flat scene `rptr_hip --dump-scene` writes is the one the Python path dumps, byte for byte"""
    exe = _build_cli(tmp_path)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vks", name)
    if not os.path.exists(src):
        pytest.skip("fixture %s not present" % name)
    py, cpp = str(tmp_path / "py.rpsc"), str(tmp_path / "cpp.rpsc")
    s = _py_dump(src, py)
    out = subprocess.run([exe, src, "--dump-scene", cpp], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    a, b = open(py, "rb").read(), open(cpp, "rb").read()
    assert len(a) == len(b)
    if a != b:
        first = next(i for i in range(len(a)) if a[i] != b[i])
        raise AssertionError("dumps differ from byte %d of %d (%d bytes differ)" % (first, len(a), sum(x != y for x, y in zip(a, b))))
    assert len(s.lights) > 0 and len(s.textures) == 3 * len(s.materials)


def test_cpp_lights_on_a_many_emitter_scene(tmp_path):
    """the bin equalisation (clones, Halton shuffles, importance padding) on 512 emitters of very different power: C++ == Python"""
    from realtimepathtracingresearchframework_amd import vks
    exe = _build_cli(tmp_path)
    s = scenes.grid(nx=60, nz=40, with_emitters=True)
    rng = np.random.default_rng(5)
    for m in s.materials:
        if m.emission_intensity > 0:
            m.emission_intensity = float(m.emission_intensity * rng.uniform(0.05, 30.0))
    s.prepare_lights()
    path = str(tmp_path / "g.vks")
    vks.write_vks(path, s)
    py, cpp = str(tmp_path / "py.rpsc"), str(tmp_path / "cpp.rpsc")
    s2 = _py_dump(path, py)
    out = subprocess.run([exe, path, "--dump-scene", cpp], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert len(s2.lights) >= 16
    assert open(py, "rb").read() == open(cpp, "rb").read()


@pytest.mark.gpu
def test_cli_renders_a_vks_file_like_the_python_host(tmp_path):
    """.vks -> C++ reader -> C++ emitter preparation -> backend, no Python in between: the validation image equals the one the Python
    host renders from the same file, bit for bit"""
    from realtimepathtracingresearchframework_amd import backend, vks
    exe = _build_cli(tmp_path)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vks", "alpha_v4.vks")
    W, H = 96, 64
    r = subprocess.run([exe, src, "--img", str(W), str(H), "--pfm", "--validation", str(tmp_path / "v"), "--validation-spp", "3"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = read_pfm(str(tmp_path / "v_0003.pfm"))
    s = vks.read_vks(src)
    rh = backend.RenderHip()
    rh.initialize(W, H)
    rh.set_scene(s)
    for k in range(3):
        rh.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=(k == 0)), spp=1)
    img = np.zeros((H, W, 4), np.float32)
    rh.readback_framebuffer(img)
    rh.close()
    assert np.array_equal(got.view(np.uint32), img[..., :3].view(np.uint32)) and got.std() > 0.01


def test_cpp_vks_reader_wide_ids_indices_and_lods(tmp_path):
    """the C++ reader on a file with 16-bit material ids, index buffers and a LoD group: same flat scene as the Python reader, for the base
    level and for --remove-first-lods 1"""
    from realtimepathtracingresearchframework_amd import vks
    exe = _build_cli(tmp_path)
    s = scenes.alpha_test()
    pm = next(i for i, p in enumerate(s.pmeshes) if p.tri_material_ids is not None)
    n = len(s.pmeshes[pm].tri_material_ids)
    wide = (np.asarray(s.pmeshes[pm].tri_material_ids, np.uint16) + np.uint16(512) * (np.arange(n) % 2).astype(np.uint16)).astype(np.uint16)
    others = [i for i in range(len(s.pmeshes)) if i != pm]
    path = str(tmp_path / "w.vks")
    vks.write_vks(path, s, wide_material_ids={pm: wide}, index_buffers=True, lod_groups=[[(others[0], 0.0), (others[1], 0.5)]])
    for lods in (0, 1):
        py, cpp = str(tmp_path / ("py%d.rpsc" % lods)), str(tmp_path / ("cpp%d.rpsc" % lods))
        vks.read_vks(path, remove_first_lods=lods).dump(py)
        out = subprocess.run([exe, path, "--dump-scene", cpp, "--remove-first-lods", str(lods)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert open(py, "rb").read() == open(cpp, "rb").read()
    assert open(str(tmp_path / "py0.rpsc"), "rb").read() != open(str(tmp_path / "py1.rpsc"), "rb").read()


def test_vks_loader_params_dynamic_keywords_and_instance_pruning(tmp_path):
    """SceneLoaderParams::PerFile (librender/scene.h:33-45) in both readers: `_SHADERMESH_` in a material's extended name flags its meshes
    dynamic (subtly dynamic with small_deformation, not at all with ignore_animation), instance_pruning_probability drops the instances
    whose halton2(index) lies below it (scene.cpp:658-706,734-749); same flat scene from Python and C++"""
    from realtimepathtracingresearchframework_amd import vks, abi
    exe = _build_cli(tmp_path)
    s = scenes.two_level_test()
    path = str(tmp_path / "f.vks")
    names = vks.write_vks(path, s)
    tex_dir = vks.texture_dir(path)
    os.makedirs(tex_dir, exist_ok=True)
    mat = next(int(p.material_offsets[0]) for p in s.pmeshes if p.tri_material_ids is None)
    with open(tex_dir + names[mat] + "_Ex.txt", "w") as f:
        f.write(names[mat] + "_SHADERMESH_wind")
    users = {i for i, p in enumerate(s.pmeshes) if p.tri_material_ids is None and mat in [int(x) for x in p.material_offsets]}
    assert users
    assert [vks.halton2(i) for i in range(4)] == [0.0, 0.5, 0.25, 0.75]
    cases = [({}, []), ({"small_deformation": True}, ["--small-deformation"]), ({"ignore_animation": True}, ["--ignore-animation"]),
             ({"instance_pruning_probability": 0.3}, ["--instance-pruning", "0.3"])]
    for k, (kw, flags) in enumerate(cases):
        r = vks.read_vks(path, **kw)
        want = 0 if "ignore_animation" in kw else abi.MESH_SUBTLY_DYNAMIC if "small_deformation" in kw else abi.MESH_DYNAMIC
        for i, p in enumerate(r.pmeshes):
            assert int(r.meshes[p.mesh].dynamic) == (want if i in users else 0)
        kept = [i for i in range(len(s.instances)) if not (kw.get("instance_pruning_probability") and vks.halton2(i) < 0.3)]
        assert len(r.instances) == len(kept) and (len(kept) == 12 or len(kept) == 8)
        py, cpp = str(tmp_path / ("py%d.rpsc" % k)), str(tmp_path / ("cpp%d.rpsc" % k))
        r.dump(py)
        out = subprocess.run([exe, path, "--dump-scene", cpp] + flags, capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert open(py, "rb").read() == open(cpp, "rb").read()


def test_cli_argument_table_follows_the_reference(tmp_path):
    """cmdline.cpp:226-259,296-494: old single-dash arguments and --benchmark-file are refused with a pointer to their successors,
    profiling automation options need --profiling, -h / --help print the usage, --profiling-frames / --frame are the old spellings of
    --profiling-fps / --keyframe (parsed; --describe shows the keyframe)"""
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    run = lambda *a: subprocess.run([exe, path] + list(a), capture_output=True, text=True)
    r = run("--validation", "x", "-spp", "4")
    assert r.returncode == 2 and "double dashes" in r.stderr and "Unknown argument: -spp" in r.stderr
    r = run("--validation", "x", "-vulkan")
    assert r.returncode == 2 and "--backend" in r.stderr
    r = run("--benchmark-file", "b.csv")
    assert r.returncode == 2 and "is now --profiling" in r.stderr
    for opt in (["--profiling-fps", "30"], ["--profiling-frames", "30"], ["--profiling-img", "p"]):
        r = run("--validation", "x", *opt)
        assert r.returncode == 2 and "without profiling mode" in r.stderr
    for h in ("-h", "--help"):
        r = run(h)
        assert r.returncode == 2 and r.stderr.startswith("usage:")
    (tmp_path / "key.ini").write_text(INI_KEY)
    r = run("--describe", "--frame", "0.25:" + str(tmp_path / "key.ini"))
    assert r.returncode == 0 and "keyframe hold 0.250000" in r.stdout
    # accepted and without effect on this backend
    r = run("--describe", "--resource-dir", "/tmp", "--deduplicate-scene", "--vulkan-device", "0", "--disable-ui")
    assert r.returncode == 0, r.stderr


@pytest.mark.gpu
def test_cli_data_capture_toggles_and_keyframes(tmp_path):
    """--data-capture over two keyframes with --data-capture-no-aovs --data-capture-normal-depth: rgba + normal_depth per keyframe, nothing
    else (cmdline.cpp:432-452, app_state.cpp:499-531); a profiling run without keyframes ends after its first frame (imstate.cpp:890-898)"""
    exe = _build_cli(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    (tmp_path / "base.ini").write_text(INI_BASE.replace("wavefront-gltf-transmission", "wavefront-gltf"))
    (tmp_path / "key.ini").write_text(INI_KEY)
    prefix = str(tmp_path / "cap")
    p = subprocess.run([exe, path, "--img", "64", "48", "--data-capture", prefix, "--data-capture-spp", "2", "--data-capture-no-aovs", "--data-capture-normal-depth",
                        "--data-capture-fps", "30", "--keyframe", str(tmp_path / "base.ini"), "--keyframe", str(tmp_path / "key.ini")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    names = sorted(f for f in os.listdir(tmp_path) if f.startswith("cap_"))
    assert names == ["cap_0001_normal_depth.exr", "cap_0001_rgba.exr", "cap_0002_normal_depth.exr", "cap_0002_rgba.exr"]
    a, b = read_exr(prefix + "_0001_rgba.exr"), read_exr(prefix + "_0002_rgba.exr")
    assert a.shape == (48, 64, 4) and not np.array_equal(a, b)          # the second keyframe moves the camera
    p = subprocess.run([exe, path, "--img", "64", "48", "--profiling", str(tmp_path / "one"), "--profiling-frames", "30"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert len((tmp_path / "one.csv").read_text().strip().splitlines()) == 2


# ---------------------------------------------------------------- rptr_compare (util/compare_exr.cpp)
def _build_compare(tmp_path):
    exe = str(tmp_path / "rptr_compare")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", os.path.join(HOST, "rptr_compare.cpp"), "-o", exe, "-lz"])
    return exe


def write_exr_py(path, planes, half=False, compression=0):
    """a scan-line OpenEXR file written independently of host/write_image.hpp: `planes` = {channel name: (h, w) array}, channels stored in
    alphabetical order as FLOAT or HALF, compression NONE (0), RLE (1), ZIPS (2) or ZIP (3) with OpenEXR's byte shuffle + delta predictor"""
    import struct
    import zlib
    names = sorted(planes)
    h, w = planes[names[0]].shape
    dt = np.float16 if half else np.float32

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<I", len(value)) + value
    ch = b"".join(n.encode() + b"\0" + struct.pack("<iBBBBii", 1 if half else 2, 0, 0, 0, 0, 1, 1) for n in names) + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    head = struct.pack("<II", 20000630, 2) + attr("channels", "chlist", ch) + attr("compression", "compression", bytes([compression])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1.0)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if compression == 3 else 1
    blocks = []
    for y0 in range(0, h, lines):
        raw = b"".join(np.ascontiguousarray(planes[n][y], dtype=dt).tobytes() for y in range(y0, min(h, y0 + lines)) for n in names)
        data = raw
        if compression:
            b = np.frombuffer(raw, np.uint8)
            t = np.concatenate([b[0::2], b[1::2]]).astype(np.int32)
            d = t.copy()
            d[1:] = (t[1:] - t[:-1] + 128 + 256) & 0xFF
            d = d.astype(np.uint8).tobytes()
            if compression == 1:      # RLE: runs of 3..128 equal bytes as (n - 1, byte), everything else as (-n, n literal bytes)
                z, i = bytearray(), 0
                while i < len(d):
                    run = 1
                    while i + run < len(d) and run < 128 and d[i + run] == d[i]:
                        run += 1
                    if run >= 3:
                        z += bytes([run - 1, d[i]])
                        i += run
                    else:
                        j = i
                        while j < len(d) and j - i < 127 and not (j + 2 < len(d) and d[j] == d[j + 1] == d[j + 2]):
                            j += 1
                        z += bytes([(256 - (j - i)) & 0xFF]) + d[i:j]
                        i = j
                z = bytes(z)
            else:
                z = zlib.compress(d)
            data = z if len(z) < len(raw) else raw
        blocks.append(struct.pack("<iI", y0, len(data)) + data)
    table_at = len(head)
    offs, at = [], table_at + 8 * len(blocks)
    for b in blocks:
        offs.append(at)
        at += len(b)
    with open(path, "wb") as f:
        f.write(head + struct.pack("<%dQ" % len(offs), *offs) + b"".join(blocks))


def test_compare_tool_follows_compare_exr(tmp_path):
    """util/compare_exr.cpp:51-97: per value |ref - cmp| / ref (|cmp| where ref is 0), above 1e-6 anywhere = not the same, exit code -1,
    the error of every value in <CMP>_err.exr; EXR files with NONE / RLE / ZIPS / ZIP compression, FLOAT and HALF channels; PFM; compressions without
    a decoder here (PXR24 ...) refused by name"""
    exe = _build_compare(tmp_path)
    rng = np.random.default_rng(3)
    H, W = 37, 29
    base = {n: rng.random((H, W)).astype(np.float32) + 0.25 for n in "RGBA"}
    base["R"][0, 0] = 0.0
    same = {n: v.copy() for n, v in base.items()}
    close = {n: v * np.float32(1.0 + 4e-7) for n, v in base.items()}
    off = {n: v.copy() for n, v in base.items()}
    off["G"][5, 7] *= np.float32(1.0 + 1e-5)
    off["R"][0, 0] = 3e-6
    for name, planes, comp in (("ref", base, 0), ("same", same, 3), ("close", close, 2), ("off", off, 3)):
        write_exr_py(str(tmp_path / (name + ".exr")), planes, compression=comp)
    run = lambda *a: subprocess.run([exe] + [str(tmp_path / x) for x in a], capture_output=True, text=True)
    r = run("ref.exr", "same.exr", "close.exr")
    assert r.returncode == 0 and "Comparing" in r.stdout, r.stderr
    r = run("ref.exr", "same.exr", "off.exr")
    assert r.returncode == 255 and "off.exr isn't the same as" in r.stderr and "same.exr isn't" not in r.stderr
    err = read_exr(str(tmp_path / "off.exr_err.exr"))
    want = np.zeros((H, W, 4), np.float32)
    want[5, 7, 1] = abs(base["G"][5, 7] - off["G"][5, 7]) / base["G"][5, 7]
    want[0, 0, 0] = np.float32(3e-6)
    assert np.array_equal(err, want)
    # half channels (what the AOV captures hold), a PFM pair, a size mismatch, an unsupported compression, the usage
    smooth = {n: np.round(v * 4).astype(np.float32) / 4 for n, v in base.items()}          # long runs after the predictor: RLE pays
    write_exr_py(str(tmp_path / "r0.exr"), smooth, compression=0)
    write_exr_py(str(tmp_path / "r1.exr"), smooth, compression=1)
    assert os.path.getsize(tmp_path / "r1.exr") < os.path.getsize(tmp_path / "r0.exr") and run("r0.exr", "r1.exr").returncode == 0
    hb = {n: v.astype(np.float16) for n, v in base.items()}
    write_exr_py(str(tmp_path / "h1.exr"), hb, half=True, compression=0)
    write_exr_py(str(tmp_path / "h2.exr"), hb, half=True, compression=3)
    assert run("h1.exr", "h2.exr").returncode == 0
    from realtimepathtracingresearchframework_amd.scenes import f32
    for name, scale in (("p1", 1.0), ("p2", 1.0), ("p3", 1.001)):
        with open(tmp_path / (name + ".pfm"), "wb") as f:
            f.write(b"PF\n%d %d\n-1.0\n" % (W, H))
            f.write(np.ascontiguousarray(np.stack([base[c] * f32(scale) for c in "RGB"], axis=-1)[::-1]).tobytes())
    assert run("p1.pfm", "p2.pfm").returncode == 0 and run("p1.pfm", "p3.pfm").returncode == 255
    write_exr_py(str(tmp_path / "small.exr"), {n: v[:10] for n, v in base.items()})
    r = run("ref.exr", "small.exr")
    assert r.returncode == 255 and "same size" in r.stderr
    raw = bytearray(open(tmp_path / "ref.exr", "rb").read())
    k = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
    raw[k] = 5
    open(tmp_path / "pxr.exr", "wb").write(raw)
    r = run("ref.exr", "pxr.exr")
    assert r.returncode == 255 and "PXR24" in r.stderr
    assert subprocess.run([exe, str(tmp_path / "ref.exr")], capture_output=True).returncode == 255


def test_compare_tool_reads_piz_compressed_validation_images(tmp_path):
    """the reference's --validation run writes PIZ-compressed OpenEXR by default (libapp/app_state.cpp:476, util/write_image.cpp:150-151),
    and its regression workflow (util/compare_exr.cpp:51-97) takes such a file as REF: bin/rptr_compare decodes them (host/read_image.hpp:
    bitmap + LUT, wavelet, Huffman with run lengths). Fixtures: tests/golden/piz/ (generator and its limits: gen_piz_fixture.py) -- FLOAT
    channels with two blocks (32 + 5 lines, odd width: the 16-bit wavelet), HALF channels in three blocks (the 14-bit wavelet), a constant
    image (runs); any further <name>.exr + <name>.f32 pair dropped there (e.g. written by a build of the reference) is checked too."""
    import glob
    exe = _build_compare(tmp_path)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "piz")
    names = sorted(glob.glob(os.path.join(gold, "*.exr")))
    assert len(names) >= 3
    for path in names:
        raw = open(path, "rb").read()
        k = raw.index(b"compression\0compression\0") + len(b"compression\0compression\0") + 4
        assert raw[k] == 4                                                       # PIZ
        want = np.fromfile(path[:-4] + ".f32", dtype="<f4")
        # the same pixels as an uncompressed file: the tool must call them equal, and a changed value different
        chans = []
        at = raw.index(b"channels\0chlist\0") + len(b"channels\0chlist\0") + 4
        while raw[at] != 0:
            e = raw.index(b"\0", at)
            chans.append((raw[at:e].decode(), int.from_bytes(raw[e + 1:e + 5], "little")))
            at = e + 17
        box = np.frombuffer(raw, "<i4", 4, raw.index(b"dataWindow\0box2i\0") + len(b"dataWindow\0box2i\0") + 4)
        W, H = int(box[2] - box[0] + 1), int(box[3] - box[1] + 1)
        planes = want.reshape(len(chans), H, W)
        half = chans[0][1] == 1
        ref = {n: (planes[i].astype(np.float16) if half else planes[i]) for i, (n, _) in enumerate(chans)}
        write_exr_py(str(tmp_path / "plain.exr"), ref, half=half, compression=0)
        r = subprocess.run([exe, path, str(tmp_path / "plain.exr")], capture_output=True, text=True)
        assert r.returncode == 0, (path, r.stderr)
        ref2 = {n: v.copy() for n, v in ref.items()}
        n0 = chans[-1][0]
        ref2[n0] = ref2[n0].copy()
        ref2[n0][H - 1, W - 1] = ref2[n0][H - 1, W - 1] * 1.01 + 0.01
        write_exr_py(str(tmp_path / "changed.exr"), ref2, half=half, compression=3)
        r = subprocess.run([exe, path, str(tmp_path / "changed.exr")], capture_output=True, text=True)
        assert r.returncode == 255 and "isn't the same" in r.stderr


def test_compare_tool_refuses_corrupt_piz_streams_without_crashing(tmp_path):
    """the PIZ decoder (host/read_image.hpp) on streams that are not what an encoder wrote -- a file cut off inside a block, a
    block whose Huffman table / bitmap / code stream has flipped bytes, a block length that points beyond the file: the tool must end with
    its error exit code (255: unreadable or different), never with a signal, never hang"""
    import glob
    exe = _build_compare(tmp_path)
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "piz")
    rng = np.random.default_rng(7)
    n_runs = 0
    for path in sorted(glob.glob(os.path.join(gold, "*.exr"))):
        raw = open(path, "rb").read()
        variants = {"cut_half": raw[:len(raw) // 2], "cut_tail": raw[:len(raw) - 7]}
        for k in range(12):   # flipped bytes at random places behind the header (offset table, block headers, tables, code streams)
            b = bytearray(raw)
            lo = min(len(b) - 1, 400)
            for _ in range(1 + k % 4):
                i = int(rng.integers(lo, len(b)))
                b[i] ^= int(rng.integers(1, 256))
            variants["flip%d" % k] = bytes(b)
        variants["zeros"] = raw[:300] + bytes(len(raw) - 300)
        for name, data in variants.items():
            bad = str(tmp_path / (name + ".exr"))
            open(bad, "wb").write(data)
            r = subprocess.run([exe, path, bad], capture_output=True, text=True, timeout=60)
            assert r.returncode in (0, 255), (path, name, r.returncode, r.stderr[-200:])   # (0: the flip hit padding or an unused table entry)
            n_runs += 1
    assert n_runs >= 40


@pytest.mark.gpu
def test_validation_images_through_the_compare_tool(tmp_path):
    """the reference's regression workflow end to end: two --validation runs of the same configuration compare equal (exit 0), a run with
    another sample count does not"""
    exe, cmp = _build_cli(tmp_path), _build_compare(tmp_path)
    s = scenes.cornell32()
    path = str(tmp_path / "c.rpsc")
    s.dump(path)
    for tag, spp in (("a", 4), ("b", 4), ("c", 5)):
        r = subprocess.run([exe, path, "--img", "96", "64", "--validation", str(tmp_path / tag), "--validation-spp", str(spp)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    assert subprocess.run([cmp, str(tmp_path / "a_0004.exr"), str(tmp_path / "b_0004.exr")], capture_output=True).returncode == 0
    r = subprocess.run([cmp, str(tmp_path / "a_0004.exr"), str(tmp_path / "c_0005.exr")], capture_output=True, text=True)
    assert r.returncode == 255 and "isn't the same" in r.stderr and os.path.exists(str(tmp_path / "c_0005.exr_err.exr"))


def test_cpp_vks_reader_keeps_mip_levels(tmp_path):
    """a .vks scene whose textures carry mip levels: the C++ reader's flat scene (levels back to back, RptrTextureDesc.mip_levels) equals the
    Python reader's byte for byte, and the dump reads back with the levels in place"""
    from realtimepathtracingresearchframework_amd import vks
    exe = _build_cli(tmp_path)
    s = scenes.textured_test()
    for t in s.textures:
        lv, cur = [], np.asarray(t.rgba)
        while cur.shape[0] > 1 or cur.shape[1] > 1:
            cur = np.ascontiguousarray(cur[::2, ::2][:max(1, cur.shape[0] // 2), :max(1, cur.shape[1] // 2)])
            lv.append(cur)
        t.mips = lv or None
    path = str(tmp_path / "t.vks")
    vks.write_vks(path, s)
    py, cpp = str(tmp_path / "py.rpsc"), str(tmp_path / "cpp.rpsc")
    r = vks.read_vks(path)
    assert any(t.mips for t in r.textures)
    r.dump(py)
    out = subprocess.run([exe, path, "--dump-scene", cpp], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert open(py, "rb").read() == open(cpp, "rb").read()
    again = str(tmp_path / "again.rpsc")
    out = subprocess.run([exe, cpp, "--dump-scene", again], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert open(again, "rb").read() == open(cpp, "rb").read()


# ---------------------------------------------------------------- a15 host half: Sun settings of an .ini refit the sky
def write_synthetic_sky_data(dirpath, seed=11):
    """data headers in the layout of the Hosek-Wilkie distribution (`double name[] = { ... };`) filled with made-up but well-behaved
    coefficients: what the parser and the fit need on a box without the published tables. The Bezier / bilinear blends are convex
    combinations, so every fitted coefficient stays in the range given per index here (finite skies, a positive sun)."""
    rng = np.random.default_rng(seed)
    rng_of = [(-1.2, -1.0), (-0.5, -0.1), (0.0, 1.0), (0.0, 1.0), (-3.0, -1.0), (0.0, 0.5), (0.0, 0.2), (0.0, 1.0), (0.3, 0.7)]

    def arr(name, values, ctype="double"):
        return "%s %s[] =\n{\n%s\n};\n\n" % (ctype, name, ",\n".join("\t%.9e" % v for v in values))

    def dataset():   # [albedo 2][turbidity 10][control point 6][coefficient 9]
        return np.concatenate([rng.uniform(lo, hi, 120)[:, None] for lo, hi in rng_of], axis=1).reshape(-1)
    rgb = "// synthetic stand-in (tests): layout of ArHosekSkyModelData_RGB.h\n"
    for c in range(3):
        rgb += arr("datasetRGB%d" % (c + 1), dataset()) + arr("datasetRGBRad%d" % (c + 1), rng.uniform(5, 20, 120))
    rgb += "double* datasetsRGB[] =\n{\n\tdatasetRGB1,\n\tdatasetRGB2,\n\tdatasetRGB3\n};\n"
    spec = "/* synthetic stand-in (tests): layout of ArHosekSkyModelData_Spectral.h */\n"
    for w in range(320, 721, 40):
        spec += arr("dataset%d" % w, dataset()) + arr("datasetRad%d" % w, rng.uniform(5, 20, 120))
    for w in range(320, 721, 40):
        coef = np.zeros((10 * 45, 4))
        coef[:, 3] = rng.uniform(1e3, 2e4, 450)      # piecewise cubics: constant term last (the loop reads backwards)
        coef[:, 2] = rng.uniform(0, 1e3, 450)
        spec += arr("solarDataset%d" % w, coef.reshape(-1))
    for w in range(320, 721, 40):
        spec += "double limbDarkeningDataset%d[] =\n{ %s };\n" % (w, ", ".join("%.6f" % v for v in [0.3, 0.9, -0.4, 0.3, -0.15, 0.05]))
    cie = "#define CM_CIE_SAMPLES 95\ntypedef float Float;\nstatic const Float cie1931_tbl[CM_CIE_SAMPLES * 3] = {\n"
    lam = np.linspace(360, 830, 95)
    tbl = np.concatenate([np.exp(-0.5 * ((lam - mu) / sg) ** 2) for mu, sg in ((600, 40), (555, 45), (450, 25))])
    cie += ", ".join("Float(%.12f)" % v for v in tbl) + "\n};\n"
    (dirpath / "sky_model_data_rgb.h").write_text(rgb)
    (dirpath / "sky_model_data_spectral.h").write_text(spec)
    (dirpath / "color_matching.h").write_text(cie)


INI_SUN = """[Application][scene.vks]
[.][Sun]
height= 2.500000e+01
angle= -6.000000e+01
turbidity= 6.500000e+00
Color= 4.000000e-01 3.000000e-01 1.000000e-01
..
"""


@pytest.mark.gpu
def test_cli_refits_the_sky_for_the_sun_settings_of_an_ini(tmp_path):
    """row a15, host half: `[.][Sun]` of a configuration file (height / angle / turbidity / Color, libapp/scene_state.h:79-96) moves the
    sun -- the host runs the Hosek-Wilkie fit (host/sky_fit.hpp) on the data headers --sky-data points at. Synthetic tables of the
    published layout here (the box has no copy of the real ones; the real ones are held against the reference's own code in
    tests/test_sky_fit.py): the CLI's image = the Python mirror's image with sky_fit.py's parameters, bit for bit; and without the data
    the scene file's sky is kept and a note says why."""
    from common import gpu_render
    from realtimepathtracingresearchframework_amd import backend, sky_fit
    exe = _build_cli(tmp_path)
    s = scenes.grid(40, 20, with_emitters=True)
    path = str(tmp_path / "g.rpsc")
    s.dump(path)
    data = tmp_path / "skydata"
    data.mkdir()
    write_synthetic_sky_data(data)
    (tmp_path / "sun.ini").write_text(INI_SUN)
    W, H, spp = 96, 64, 2
    p = subprocess.run([exe, path, "--validation", str(tmp_path / "sun"), "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--variant", "gltf",
                        "--config", str(tmp_path / "sun.ini"), "--sky-data", str(data)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    moved = read_pfm("%s_%04d.pfm" % (tmp_path / "sun", spp))
    tables = sky_fit.SkyTables(str(data))
    assert tables.has_sun
    sp = sky_fit.fit_sky(tables, sky_fit.sun_dir_from_height_angle(25.0, -60.0), 6.5, (0.4, 0.3, 0.1), len(s.lights))
    assert sp.sun_radiance[0] > 0 and sp.sun_radiance[3] == 0.5 and np.isfinite(np.array([list(r) for r in sp.sky_params.configs])).all()
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    r.update_config(sp)
    r.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=spp)
    ref = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(ref)
    r.close()
    assert np.isfinite(moved).all() and np.array_equal(moved.view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32))
    # the environment variable does what the flag does; without the data the file's sky stays and the host says so
    p = subprocess.run([exe, path, "--validation", str(tmp_path / "env"), "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--variant", "gltf",
                        "--config", str(tmp_path / "sun.ini")], capture_output=True, text=True, env=dict(os.environ, RPTR_SKY_DATA=str(data)))
    assert p.returncode == 0, p.stderr
    assert np.array_equal(read_pfm("%s_%04d.pfm" % (tmp_path / "env", spp)).view(np.uint32), moved.view(np.uint32))
    env = {k: v for k, v in os.environ.items() if k != "RPTR_SKY_DATA"}
    p = subprocess.run([exe, path, "--validation", str(tmp_path / "kept"), "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--variant", "gltf",
                        "--config", str(tmp_path / "sun.ini")], capture_output=True, text=True, env=env)
    assert p.returncode == 0 and "--sky-data" in p.stderr
    kept = read_pfm("%s_%04d.pfm" % (tmp_path / "kept", spp))
    plain, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    assert np.array_equal(kept.view(np.uint32), np.ascontiguousarray(plain[..., :3]).view(np.uint32)) and not np.array_equal(kept, moved)
    # a directory without the headers is an error, not a silent fallback
    p = subprocess.run([exe, path, "--validation", str(tmp_path / "bad"), "--img", "8", "8", "--sky-data", str(tmp_path)], capture_output=True, text=True)
    assert p.returncode == 3 and "sky_model_data_rgb.h" in p.stderr


def test_synthetic_sky_tables_parse_like_the_real_layout(tmp_path):
    """(CPU) the stand-in headers go through the same parsers (C++ and Python) and give the same fit in both"""
    import ctypes as C
    from test_sky_fit import as_words, build_shim
    from realtimepathtracingresearchframework_amd import sky_fit
    write_synthetic_sky_data(tmp_path)
    t = sky_fit.SkyTables(str(tmp_path))
    assert t.has_sun and [len(v) for v in t.rgb] == [1080] * 3 and [len(v) for v in t.solar] == [1800] * 11 and len(t.cie) == 285
    L = build_shim(tmp_path)
    err = C.create_string_buffer(512)
    for d, turb, al, lights in (((0.3, 0.8, 0.5), 3.0, (0.2, 0.2, 0.2), 0), ((0.8, 0.25, 0.3), 6.5, (0.4, 0.3, 0.1), 4), ((0.0, -1.0, 0.0), 10.0, (1.0, 0.0, 0.5), 0)):
        got = abi.SceneParams()
        assert L.shim_fit(str(tmp_path).encode(), (C.c_float * 3)(*d), C.c_float(turb), (C.c_float * 3)(*al), lights, C.byref(got), err, 512) == 0, err.value
        assert np.array_equal(as_words(got), as_words(sky_fit.fit_sky(t, d, turb, al, lights)))


# ---------------------------------------------------------------- several scene files per run, --deduplicate-scene (SURVEY 8f rank 1 / 2 remainder)
def _describe(exe, *args):
    out = subprocess.run([exe, *args, "--describe"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    words = [l for l in out.stdout.strip().splitlines() if l.startswith("geometries ")][0].split()
    return dict(zip(words[0::2], words[1::2])), out.stdout


def test_cli_appends_further_scene_files_and_deduplicates(tmp_path):
    """`<scene_file> [<scene_file>...]` (librender/scene.cpp:50-69): the files' meshes / parameterized meshes / instances / materials /
    textures are appended with shifted indices, the emitters binned over the whole scene; --deduplicate-scene (Scene::deduplicate +
    garbage_collect, :142-148) merges equal meshes, materials and textures and drops what nothing refers to any more. No GPU needed."""
    exe = _build_cli(tmp_path)
    a, b = scenes.textured_test(), scenes.grid(20, 10, with_emitters=True)
    pa, pb = str(tmp_path / "a.rpsc"), str(tmp_path / "b.rpsc")
    a.dump(pa)
    b.dump(pb)
    ka, _ = _describe(exe, pa)
    kb, _ = _describe(exe, pb)
    kab, _ = _describe(exe, pa, pb)
    for k in ("geometries", "meshes", "parameterized_meshes", "instances", "materials", "triangles", "qsum"):
        assert int(kab[k]) == int(ka[k]) + int(kb[k]), k
    merged = scenes.textured_test().append(scenes.grid(20, 10, with_emitters=True))
    assert int(kab["lights"]) == len(merged.lights) and len(merged.lights) > len(b.lights)      # binned over all emitters of both files
    assert abs(float(kab["fovy"]) - a.camera_params().fovy) < 1e-5                               # the first file's camera
    # the same file twice: everything doubles; de-duplicated, the second copy's meshes, materials and textures fold into the first's
    kaa, _ = _describe(exe, pa, pa)
    kd, text = _describe(exe, pa, pa, "--deduplicate-scene")
    assert int(kaa["meshes"]) == 2 * int(ka["meshes"]) and int(kaa["instances"]) == 2 * int(ka["instances"])
    assert int(kd["meshes"]) == int(ka["meshes"]) and int(kd["geometries"]) == int(ka["geometries"]) and int(kd["triangles"]) == int(ka["triangles"])
    assert int(kd["instances"]) == 2 * int(ka["instances"]) and int(kd["parameterized_meshes"]) == 2 * int(ka["parameterized_meshes"])
    assert int(kd["materials"]) <= int(kaa["materials"]) and "Duplicate geometry detected" in text
    # the merged scene survives a round trip through the flat layout
    out = str(tmp_path / "ab.rpsc")
    assert subprocess.run([exe, pa, pb, "--dump-scene", out], capture_output=True).returncode == 0
    kr, _ = _describe(exe, out)
    assert {k: kr[k] for k in ("geometries", "meshes", "instances", "materials", "lights", "triangles", "qsum")} == \
        {k: kab[k] for k in ("geometries", "meshes", "instances", "materials", "lights", "triangles", "qsum")}


@pytest.mark.gpu
def test_cli_renders_several_scene_files_like_the_merged_scene(tmp_path):
    """the image of `a.rpsc b.rpsc` = the image of the Python host's merged scene (Scene.append), bit for bit, and the same with
    --deduplicate-scene on a scene that holds the same meshes twice (textures and their handles survive the re-indexing)"""
    from common import gpu_render
    exe = _build_cli(tmp_path)
    a, b = scenes.textured_test(), scenes.grid(20, 10, with_emitters=True)
    pa, pb = str(tmp_path / "a.rpsc"), str(tmp_path / "b.rpsc")
    a.dump(pa)
    b.dump(pb)
    W, H, spp = 96, 64, 2

    def cli(prefix, *files):
        p = subprocess.run([exe, *files, "--validation", str(tmp_path / prefix), "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--variant", "gltf"],
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        return read_pfm("%s_%04d.pfm" % (tmp_path / prefix, spp))
    merged = scenes.textured_test().append(scenes.grid(20, 10, with_emitters=True))
    ref, _, _ = gpu_render(merged, W, H, spp, abi.VARIANT_GLTF)
    got = cli("ab", pa, pb)
    assert np.array_equal(got.view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32))
    assert not np.array_equal(got, cli("a", pa))                       # the second file is in the picture
    # de-duplication: same image. (With a literal emission colour: collect_emitters multiplies the raw base-colour words, librender/lights.cpp:14-73,
    # so for an emitter whose colour is a texture HANDLE the light's radiance changes with the texture's index -- in the reference as here.)
    c = scenes.textured_test()
    for m in c.materials:
        if m.emission_intensity > 0:
            m.base_color[:] = [1.0, 0.8, 0.6]
    c.prepare_lights()
    pc = str(tmp_path / "c.rpsc")
    c.dump(pc)
    twice = cli("cc", pc, pc)
    assert np.array_equal(twice.view(np.uint32), cli("ccd", pc, pc, "--deduplicate-scene").view(np.uint32))
    assert not np.array_equal(twice, cli("c", pc))


@pytest.mark.gpu
def test_cli_profiling_reproduces_the_benchmarks_schedule(tmp_path):
    """the C++ host gets the schedule bench.py measures -- 11 frame contexts x 4 frames per launch sequence, a hardware queue
    per context (bin/rptr_hip passes RPTR_CREATE_SET_HW_QUEUES: its first create sets GPU_MAX_HW_QUEUES when nobody did, csrc/host_state.h ensure_hw_queues) -- and with it bench.py's
    ms per frame on C2 (1 M triangles, 1080p, 4 spp, diffuse), within a few percent."""
    import json
    import re
    import sys
    exe = _build_cli(tmp_path)
    s = scenes.grid_1m()
    path = str(tmp_path / "grid1m.rpsc")
    s.dump(path)
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}      # nobody sets it for the host: the library must
    p = subprocess.run([exe, path, "--profiling", str(tmp_path / "prof"), "--profiling-count", "200", "--frames-in-flight", "11", "--frames-per-launch", "4",
                        "--img", "1920", "1080", "--batch-spp", "4", "--variant", "diffuse"], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr
    assert "hardware queues" not in p.stderr                                     # (the warning of a host that initialised HIP with too few)
    cli_ms = float(re.search(r"([0-9.]+) ms per frame \(wall\)", p.stdout).group(1))
    rows = open(str(tmp_path / "prof.csv")).read().strip().splitlines()
    assert rows[0] == "frames_total,keyframe,frames_accumulated,render_time_ms,app_time_ms" and len(rows) == 201
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "200"], capture_output=True, text=True, cwd=ROOT)
    assert b.returncode == 0, b.stderr[-2000:]
    bench_ms = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"]
    print("C2 ms per frame: bin/rptr_hip --profiling %.4f, bench.py %.4f (ratio %.3f)" % (cli_ms, bench_ms, cli_ms / bench_ms))
    assert abs(cli_ms / bench_ms - 1.0) < 0.05
    # the same with the camera MOVING every frame (bench.py's path: --fly-through; a camera per frame inside a launch
    # sequence, rptr_hip_render_batch_cameras_async) -- and bench.py's own `boundary` leg reports the drop-in's figures beside `value`
    p = subprocess.run([exe, path, "--profiling", str(tmp_path / "prof2"), "--profiling-count", "200", "--frames-in-flight", "11", "--frames-per-launch", "4",
                        "--img", "1920", "1080", "--batch-spp", "4", "--variant", "diffuse", "--fly-through"], capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stderr
    assert "a camera per frame" in p.stdout
    fly_ms = float(re.search(r"([0-9.]+) ms per frame \(wall\)", p.stdout).group(1))
    print("... with the fly-through: %.4f (ratio %.3f)" % (fly_ms, fly_ms / bench_ms))
    assert abs(fly_ms / bench_ms - 1.0) < 0.05
    bd = json.loads([l for l in b.stdout.splitlines() if l.startswith("{")][-1]).get("boundary")
    # (its deep-queue figure is not held against bench_ms HERE: under pytest three processes -- this one, bench.py, its child -- hold HIP
    # queues, more than the GPU has, and the driver time-slices them: 3.6 ms per frame where the same command run from a shell gives 1.135
    # next to a value of 1.123, profiles/r05_notes.md; the two direct runs above are the comparison)
    assert bd and bd["full_schedule"]["frames"] >= 400 and bd["swap_buffers_2"]["ms_per_frame"] < bd["synchronous"]["ms_per_frame"]
    print("bench.py boundary leg (nested under pytest):", bd)


def _partitioned_scene():
    """one object exported in three partitions (three meshes of one material each under the same transform) + an unrelated instance"""
    s = scenes.Scene(name="partitions")
    s.materials = [abi.make_material((0.8, 0.3, 0.2), roughness=0.6), abi.make_material((0.2, 0.7, 0.3), roughness=0.3, metallic=1.0),
                   abi.make_material((0.3, 0.3, 0.9), roughness=0.8), abi.make_material((0.7, 0.7, 0.7), roughness=0.9)]
    M = np.array([[0.9, 0, 0.2, 0.1], [0, 1.1, 0, 0.0], [-0.2, 0, 0.9, -0.3]], np.float32)
    for k in range(3):
        P, N, UV = scenes._heightfield(10, 10, -1.5 + k, -0.5 + k, -1.0, 1.0, lambda X, Z: 0.25 * np.sin(3 * X) * np.cos(2 * Z))
        mesh = scenes._add_mesh(s, P, N, UV)
        s.pmeshes.append(scenes.ParameterizedMesh(mesh=mesh, material_offsets=np.array([k], np.int32)))
        s.instances.append(scenes.Instance(transform=M.copy(), pmesh=k))
    P, N, UV = scenes._heightfield(6, 6, -4.0, 4.0, -4.0, 4.0, lambda X, Z: -0.6 + 0.0 * X)
    g = scenes._add_mesh(s, P, N, UV)
    s.pmeshes.append(scenes.ParameterizedMesh(mesh=g, material_offsets=np.array([3], np.int32)))
    s.instances.append(scenes.Instance(transform=scenes.IDENTITY.copy(), pmesh=3))
    s.camera = dict(eye=(0, 2.0, 5.0), center=(0, 0, 0), up=(0, 1, 0), fov=45.0)
    s.config = scenes.SceneConfig(**scenes.SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s


def test_cli_merges_partition_instances(tmp_path):
    """--merge-partition-instances (SceneLoaderParams::PerFile::merge_partition_instances, librender/scene.cpp:757-797): consecutive instances
    with one transform become one instance whose mesh lists all their geometries; same triangles, fewer instances. No GPU needed."""
    exe = _build_cli(tmp_path)
    s = _partitioned_scene()
    path = str(tmp_path / "p.rpsc")
    s.dump(path)
    k0, _ = _describe(exe, path)
    k1, text = _describe(exe, path, "--merge-partition-instances")
    assert (int(k0["instances"]), int(k1["instances"])) == (4, 2) and "merged 2 partition instances" in text
    # (the merged-away meshes stay in the tables, unreferenced by any instance, as in the reference: their two geometries are now listed by
    # mesh 0 as well)
    assert int(k1["geometries"]) == int(k0["geometries"]) + 2 and int(k1["triangles"]) == int(k0["triangles"]) + 2 * 200 and int(k1["meshes"]) == int(k0["meshes"])
    twin = _partitioned_scene()
    assert twin.merge_partition_instances() == 2 and len(twin.instances) == 2 and twin.meshes[0].num_geometries == 3
    assert list(twin.pmeshes[0].material_offsets) == [0, 1, 2]


@pytest.mark.gpu
def test_merged_partition_instances_render_the_same_image(tmp_path):
    from common import gpu_render
    exe = _build_cli(tmp_path)
    s = _partitioned_scene()
    path = str(tmp_path / "p.rpsc")
    s.dump(path)
    W, H, spp = 96, 64, 2
    imgs = []
    for extra in ([], ["--merge-partition-instances"]):
        prefix = str(tmp_path / ("m%d" % len(extra)))
        p = subprocess.run([exe, path, "--validation", prefix, "--validation-spp", str(spp), "--img", str(W), str(H), "--pfm", "--variant", "gltf"] + extra,
                           capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        imgs.append(read_pfm("%s_%04d.pfm" % (prefix, spp)))
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))       # no coincident surfaces: ids decide nothing
    twin = _partitioned_scene()
    twin.merge_partition_instances()
    ref, _, _ = gpu_render(twin, W, H, spp, abi.VARIANT_GLTF)
    assert np.array_equal(imgs[1].view(np.uint32), np.ascontiguousarray(ref[..., :3]).view(np.uint32))
