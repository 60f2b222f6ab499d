"""BASELINE.json configs[2] (C3) and configs[4] (C5) at FULL size against the oracle, the RGBA8 half of process_samples, and the
flattened C4 tree against the oracle walking the very same tree.

Sizes: C3 = grid_1m_lights (1 000 512 triangles, 512 of them emissive), glTF BSDF + binned-RIS NEE, 1920x1080, 8 spp;
C5 = the 1 M-triangle height field as a dynamic mesh, 3840x2160, 2 spp, device-side vertex updates + refit per frame.
The oracle renders bands of rows of the same full-size frames (same pixels, same seeds): RMSE < 1e-3 (north_star).
"""
import numpy as np
import pytest

import oracle_lib as O
from common import RMSE_TOL, assert_ray_visit_parity, gpu_render, image_error, random_queries
from realtimepathtracingresearchframework_amd import abi, backend, scenes

pytestmark = pytest.mark.gpu


# ---------------------------------------------------------------- C3
def test_c3_full_size_gltf_area_lights_8spp():
    s = scenes.grid_1m_lights()
    assert s.num_tris() == 1_000_512 and len(s.lights) >= 512
    W, H, spp = 1920, 1080, 8
    a, sa, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True)
    assert np.isfinite(a).all() and sa.spp == spp
    # ray-count sanity: one primary ray per pixel sample, at most max_path_depth closest-hit queries per path, at most one shadow
    # query per closest hit; the emitters matter (shadow rays towards them: more than the sun-only frame would issue)
    n = W * H * spp
    assert n <= sa.raw.rays_closest <= 9 * n and sa.raw.hits_shaded < sa.raw.rays_closest
    assert 0 < sa.raw.rays_shadow <= sa.raw.hits_shaded
    # every ray of a (small) frame of this scene walks the exported tree like the oracle: results and visit counts, ray by ray
    osc = O.OracleScene(s)
    assert_ray_visit_parity(r, osc, 96, 54, 1, abi.VARIANT_GLTF)
    r.close()
    # determinism (same seed needs a fresh handle: a reset on the same handle advances frame_offset)
    b, sb, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.raw.rays_closest == sb.raw.rays_closest and sa.raw.rays_shadow == sb.raw.rays_shadow
    # bands of rows of the full-size frame against the oracle: sky/horizon rows, the far field, the near field
    osc.build_bvh()
    for rows in ((300, 304), (560, 564), (900, 904)):
        ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, rows=rows)
        rmse, same, maxabs = image_error(a[rows[0]:rows[1]], ref[rows[0]:rows[1]])
        assert same and rmse < RMSE_TOL, (rows, rmse, maxabs)
        assert np.array_equal(a[rows[0]:rows[1], :, 3], ref[rows[0]:rows[1], :, 3])


# ---------------------------------------------------------------- C5
def _device_buffer(xyz):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(xyz, dtype=np.float32)).cuda()
    torch.cuda.synchronize()
    return t


def test_c5_full_size_animated_4k_refit_per_frame():
    NX, NZ = 1000, 500
    s = scenes.grid(NX, NZ, deform_t=0.0, name="grid-1M-dynamic")
    assert s.num_tris() == 1_000_000
    W, H, spp = 3840, 2160, 2
    times = [1 / 60, 2 / 60, 3 / 60, 0.4]
    bufs = [_device_buffer(scenes.grid_positions(NX, NZ, t)) for t in times]
    cam = s.camera_params()

    def run(fif):
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(s)
        images, queue = [], []

        def collect():
            st = r.wait(queue.pop(0))
            assert st.spp == spp
            img = np.zeros((H, W, 4), np.float32)
            assert r.readback_framebuffer(img) == W * H * 4
            images.append(img)
        for k, buf in enumerate(bufs):
            r.update_vertices_device(0, buf.data_ptr(), buf.shape[0])
            r.refit()
            cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
            queue.append(r.render_async(cfg, spp=spp))
            if len(queue) >= fif:
                collect()
        while queue:
            collect()
        return r, images

    r1, one = run(1)
    # refit-then-trace == oracle rebuild-then-trace, on the geometry of the last frame
    osc = O.OracleScene(s)
    osc.set_dynamic_vertices(0, scenes.grid_positions(NX, NZ, times[-1]))
    osc.build_bvh()
    q = random_queries(np.random.default_rng(21), 1 << 17, -60, 60)
    q[:, 1] = np.abs(q[:, 1]) * 0.2 + 3.0
    q[:, 5] = -np.abs(q[:, 5])
    res = r1.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_OWN, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).mean() > 0.2
    r1.close()
    # band of the last full-size frame against the oracle: the 4th reset of this handle -> frame_offset = 3 frames x 2 samples
    last = one[-1]
    assert np.isfinite(last).all()
    for rows in ((1100, 1104), (1800, 1804)):
        ref_img, _ = osc.render(W, H, spp, variant=abi.VARIANT_SIMPLE, rows=rows, frame_offset=(len(times) - 1) * spp)
        rmse, same, maxabs = image_error(last[rows[0]:rows[1]], ref_img[rows[0]:rows[1]])
        assert same and rmse < RMSE_TOL, (rows, rmse, maxabs)
    # frames in flight == one frame at a time, every frame, bit for bit
    r3, three = run(3)
    r3.close()
    assert len(three) == len(one) == len(times)
    for x, y in zip(three, one):
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32))
    assert not np.array_equal(one[-1], one[-2])    # the surface moved between the frames


# ---------------------------------------------------------------- a18: the RGBA8 frame buffer against process_samples.comp
def _u8_close(got, want):
    """RGBA8 values agree up to one code in a few pixels: powf of the device vs libm at a rounding boundary"""
    d = np.abs(got.astype(np.int16) - want.astype(np.int16))
    return int(d.max()) <= 1 and float((d > 0).mean()) < 2e-3


@pytest.mark.parametrize("exposure,tonemap", [(0.0, -1), (1.5, -1), (-2.0, 0), (0.5, 1), (2.0, 2)])
def test_rgba8_framebuffer_matches_process_samples(exposure, tonemap):
    s = scenes.grid(120, 60, with_emitters=True)
    W, H = 240, 136
    params = abi.RenderParams.default()
    params.exposure = exposure
    params.early_tone_mapping_mode = tonemap
    img, _, r = gpu_render(s, W, H, 3, abi.VARIANT_GLTF, params=params, keep=True)
    u8 = np.zeros((H, W, 4), np.uint8)
    assert r.readback_framebuffer(u8) == W * H * 4
    r.close()
    want = O.process_samples_u8(img, params)
    assert _u8_close(u8, want)
    assert u8[..., :3].std() > 5                      # an image, not a constant
    if exposure == 0.0 and tonemap < 0:
        assert np.array_equal(want, O.resolve_u8(img, 0.0))
    # ... and against the oracle's own frame (float parity carries over to the 8-bit view up to one code at boundaries)
    ref, _ = O.OracleScene(s).render(W, H, 3, variant=abi.VARIANT_GLTF)
    ref_u8 = O.process_samples_u8(ref, params)
    d = np.abs(u8.astype(np.int16) - ref_u8.astype(np.int16))
    assert d.max() <= 2 and (d > 0).mean() < 0.02


@pytest.mark.parametrize("channel,moment", [(1, 0), (1, 1), (2, 0), (2, 1), (3, 0), (3, 1)])
def test_rgba8_output_channel_views(channel, moment):
    """OUTPUT_CHANNEL_ALBEDO_ROUGHNESS / NORMAL_DEPTH / MOTION_JITTER (process_samples.comp:150-178): the frame buffer shows the AOV
    images; exposure does not apply to them (:143-144)"""
    s = scenes.textured_test()
    W, H = 160, 120
    params = abi.RenderParams.default()
    params.output_channel, params.output_moment, params.exposure = channel, moment, 1.0
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    r.params = params
    cam0 = s.camera_params()
    cam1 = s.camera_params()
    cam1.pos[0] += 0.2                           # a moving camera: motion vectors are not all zero
    for cam, reset in ((cam0, True), (cam1, True)):
        r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=reset), spp=1)
    img = np.zeros((H, W, 4), np.float32)
    u8 = np.zeros((H, W, 4), np.uint8)
    assert r.readback_framebuffer(img) and r.readback_framebuffer(u8)
    aovs = [np.zeros((H, W, 4), np.float16) for _ in range(3)]
    for k in range(3):
        assert r.readback_aov(k, aovs[k]) == W * H * 4
    r.close()
    want = O.process_samples_u8(img, params, cam_pos=tuple(cam1.pos), aovs=aovs)
    assert _u8_close(u8, want)
    assert u8[..., :3].std() > 2


def test_rgba8_upscale_factor_2_replicates_pixels():
    s = scenes.cornell32()
    W, H = 96, 64
    params = abi.RenderParams.default()
    params.render_upscale_factor = 2
    img, _, r = gpu_render(s, W, H, 2, abi.VARIANT_GLTF, params=params, keep=True)
    big = np.zeros((2 * H, 2 * W, 4), np.uint8)
    assert r.readback_framebuffer(big) == W * H * 4          # element count of the RENDER resolution (get_framebuffer_size)
    with pytest.raises(backend.BackendError):
        r._check(r._L.rptr_hip_readback_u8(r._h, big.ctypes.data, W * H * 4))   # too small for the upscaled frame buffer
    r.close()
    assert _u8_close(big, O.process_samples_u8(img, params))
    assert np.array_equal(big[0::2, 0::2], big[1::2, 1::2])


# ---------------------------------------------------------------- read-backs and frames in flight
def test_readback_after_resubmitting_on_the_same_context_is_an_error():
    """wait(t0) -> render_async (reuses t0's context) -> read-back: the image of t0 is being overwritten; the call fails instead of
    returning a torn image. Reading back BEFORE the submission, or after waiting for the newer frame, works."""
    s = scenes.cornell32()
    W = H = 64
    r = backend.RenderHip(frames_in_flight=2)
    r.initialize(W, H)
    r.set_scene(s)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    t0 = r.render_async(cfg, spp=1)
    t1 = r.render_async(cfg, spp=1)
    r.wait(t0)
    first = np.zeros((H, W, 4), np.float32)
    assert r.readback_framebuffer(first) == W * H * 4           # fine: nothing newer on t0's context yet
    t2 = r.render_async(cfg, spp=1)                              # lands on t0's context
    buf = np.zeros((H, W, 4), np.float32)
    with pytest.raises(backend.BackendError):
        r.readback_framebuffer(buf)
    with pytest.raises(backend.BackendError):
        r.readback_aov(0, np.zeros((H, W, 4), np.float16))
    r.wait(t1)
    assert r.readback_framebuffer(buf) == W * H * 4             # t1's image: its context is untouched
    r.wait(t2)
    assert r.readback_framebuffer(buf) == W * H * 4
    assert not np.array_equal(buf, first)                       # three resets, three seeds
    r.close()


# ---------------------------------------------------------------- the library's own gather (csrc/host_comm.h)
@pytest.mark.parametrize("transport", ["copy", "peer"])
@pytest.mark.parametrize("world,fif,stripe_rows", [(2, 1, 32), (3, 3, 8), (8, 2, 8)])
def test_gather_all_of_n_handles_on_one_device_is_bit_identical_to_one_rank(world, fif, stripe_rows, transport, monkeypatch):
    """one process, `world` handles (all on device 0: the rows travel by device copies, the transport of rigs whose ranks share a
    device; stream / event structure and the assembly kernel are those of the RCCL transport -- or, transport "peer", every rank
    writes its rows straight into rank 0's frame from a kernel of its own, no receive buffer and no assembly pass): a sequence of
    frames, every rank rendering its stripes with `fif` frames in flight, gathered by rptr_hip_gather_all -> the assembled frames
    equal the frames of ONE rank rendering everything, bit for bit, frame by frame"""
    monkeypatch.setenv("RPTR_COMM_TRANSPORT", transport)
    s = scenes.grid(120, 60, with_emitters=True)
    W, H, spp, n_frames = 200, 120, 2, 5
    cam = s.camera_params()

    def cfg(k):
        return backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=(k % 2 == 0))

    ref = backend.RenderHip()
    ref.initialize(W, H)
    ref.set_scene(s)
    want = []
    for k in range(n_frames):
        ref.render(cfg(k), spp=spp)
        img = np.zeros((H, W, 4), np.float32)
        ref.readback_framebuffer(img)
        want.append(img)
    ref.close()

    rs = [backend.RenderHip(rank=k, world_size=world, stripe_rows=stripe_rows, frames_in_flight=fif) for k in range(world)]
    for r in rs:
        r.initialize(W, H)
        r.set_scene(s)
    with pytest.raises(backend.BackendError):
        rs[0].gather()                                   # no communicator yet
    assert rs[0].comm_transport() is None
    backend.RenderHip.comm_init_all(rs)
    assert all(r.comm_transport() == transport for r in rs)
    with pytest.raises(backend.BackendError):
        rs[0].gather()                                   # member of a one-process group: gather_all
    got, queue = [], []

    def collect():
        tickets = queue.pop(0)
        for r, t in zip(rs, tickets):
            r.wait(t)
        backend.RenderHip.gather_all(rs)
        img = np.zeros((H, W, 4), np.float32)
        assert rs[0].readback_gathered(img) == W * H * 4
        got.append(img)
    for k in range(n_frames):
        queue.append([r.render_async(cfg(k), spp=spp) for r in rs])
        if len(queue) >= fif:
            collect()
    while queue:
        collect()
    n, ms = rs[0].comm_stats()
    assert n == n_frames and ms > 0.0
    with pytest.raises(backend.BackendError):
        rs[1].gathered_frame_ptr()
    for r in rs:
        r.close()
    for a, b in zip(got, want):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_rccl_gather_on_one_gpu_through_self_send(monkeypatch):
    """The RCCL transport on the one GPU of this box: world 1, and RPTR_COMM_SELF=1 routes rank 0's own rows through
    ncclSend / ncclRecv (peer 0) and the receive buffer -- librccl is loaded at run time, a communicator is made from a unique id,
    grouped point-to-point runs on the communication stream, the assembly kernel reads the received rows. (Transfers between
    different GPUs need the multi-GPU node: bench.py --gpus N.)"""
    monkeypatch.setenv("RPTR_COMM_SELF", "1")
    s = scenes.cornell32()
    W, H = 160, 96
    r = backend.RenderHip(frames_in_flight=2)
    r.initialize(W, H)
    r.set_scene(s)
    uid = backend.RenderHip.comm_unique_id()
    assert len(uid) == abi.COMM_ID_BYTES and any(uid)
    r.comm_init_rank(uid)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    t0 = r.render_async(cfg, spp=2)
    t1 = r.render_async(cfg, spp=2)
    images = []
    for t in (t0, t1):
        r.wait(t)
        own = np.zeros((H, W, 4), np.float32)
        r.readback_framebuffer(own)
        r.gather()
        got = np.zeros((H, W, 4), np.float32)
        assert r.readback_gathered(got) == W * H * 4
        assert np.array_equal(got.view(np.uint32), own.view(np.uint32))
        images.append(got)
    assert not np.array_equal(images[0], images[1])
    t2 = r.render_async(cfg, spp=1)          # reuses t0's context: waits for that context's send on the device
    r.wait(t2)
    r.gather()
    assert r.comm_stats()[0] == 3
    r.initialize(W, H)                       # a resize drops the communicator with its frame-sized buffers
    with pytest.raises(backend.BackendError):
        r.gather()
    r.close()


# ---------------------------------------------------------------- the transmission lobe
def test_transmission_variant_image_parity_on_a_glass_scene():
    """RPTR_VARIANT_GLTF_TRANSMISSION (gltf_bsdf.glsl with GLTF_SUPPORT_TRANSMISSION[_ROUGHNESS]): solid glass (ONESIDED, refraction),
    thin frosted glass and a clear pane; image within tolerance of the oracle, ray counts equal up to branch flips, the lobe matters, and
    a scene without transmissive materials renders bit-identically in both builds of the BSDF"""
    s = scenes.glass_test()
    W, H, spp = 192, 192, 4
    img, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF_TRANSMISSION, keep=True)
    osc = O.OracleScene(s)
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF_TRANSMISSION)
    rmse, same, maxabs = image_error(img, ref)
    assert same and rmse < RMSE_TOL, (rmse, maxabs)
    assert abs(int(st.raw.rays_closest) - int(ost.rays_closest)) <= max(8, 2e-3 * ost.rays_closest)
    assert abs(int(st.raw.rays_shadow) - int(ost.rays_shadow)) <= max(8, 2e-3 * ost.rays_shadow)
    plain, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, renderer=r, reset=True)
    r.close()
    assert image_error(img, plain)[0] > 0.02           # light passes through the glass
    assert st.raw.rays_closest > 1.05 * W * H * spp
    # near-specular glass: GGX lobes of alpha 0.002 .. 0.0025 amplify a last-bit difference of the device's sin / cos against libm's by
    # 1 / alpha; most pixels are still bit-identical, a few paths flip -- the image is judged by how many pixels differ, not by RMSE
    clear = scenes.glass_test(clear=True)
    img, st, _ = gpu_render(clear, W, H, spp, abi.VARIANT_GLTF_TRANSMISSION)
    ref, ost = O.OracleScene(clear).render(W, H, spp, variant=abi.VARIANT_GLTF_TRANSMISSION)
    d = np.abs(img[..., :3] - ref[..., :3]).max(axis=2)
    assert np.isfinite(img).all() and float(np.median(d)) == 0.0 and (d > 1e-3).mean() < 0.01
    assert abs(int(st.raw.rays_closest) - int(ost.rays_closest)) <= 2e-3 * ost.rays_closest
    assert abs(float(img[..., :3].mean()) - float(ref[..., :3].mean())) < 2e-3 * float(ref[..., :3].mean())
    opaque = scenes.cornell32()
    a, _, _ = gpu_render(opaque, 96, 96, 2, abi.VARIANT_GLTF_TRANSMISSION)
    b, _, _ = gpu_render(opaque, 96, 96, 2, abi.VARIANT_GLTF)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


# ---------------------------------------------------------------- several frames in one launch sequence
@pytest.mark.parametrize("reset_rest", [True, False])
def test_batched_frames_are_bit_identical_to_frames_rendered_one_by_one(reset_rest):
    """rptr_hip_render_batch_async: 3 frames x 2 spp share their launches; every frame keeps its frame_offset / sample indices, its
    image and its ticket. reset_rest: every frame restarts the accumulation (the benchmark's pattern) / the frames accumulate
    progressively. Images, spp and ray totals equal those of the same frames submitted one by one, with a following unbatched frame too."""
    s = scenes.grid(120, 60, with_emitters=True)
    W, H, spp = 160, 96, 2
    cam = s.camera_params()

    def run(batched):
        r = backend.RenderHip(frames_in_flight=2)
        r.initialize(W, H)
        r.set_scene(s)
        images, spps, rays = [], [], 0

        def collect(t):
            nonlocal rays
            st = r.wait(t)
            img = np.zeros((H, W, 4), np.float32)
            assert r.readback_framebuffer(img) == W * H * 4
            u8 = np.zeros((H, W, 4), np.uint8)
            assert r.readback_framebuffer(u8) == W * H * 4
            images.append((img, u8))
            spps.append(st.spp)
            rays += st.raw.rays_closest + st.raw.rays_shadow
        first = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
        collect(r.render_async(first, spp=1))                        # something accumulated before the batch: frame_id = 1
        cfg0 = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=reset_rest)
        if batched:
            tickets = r.render_batch_async(cfg0, spp=spp, n_frames=3, reset_rest=reset_rest)
            assert tickets == [tickets[0], tickets[0] + 1, tickets[0] + 2]
            for t in (tickets[1], tickets[0], tickets[2]):           # any order; the middle frame first
                collect(t)
            images[1], images[2] = images[2], images[1]
            spps[1], spps[2] = spps[2], spps[1]
            with pytest.raises(backend.BackendError):
                r.wait(tickets[1])                                   # waited for already
        else:
            for k in range(3):
                collect(r.render_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=reset_rest), spp=spp))
        collect(r.render_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=False), spp=1))
        r.close()
        return images, spps, rays

    ref_images, ref_spps, ref_rays = run(False)
    images, spps, rays = run(True)
    assert spps == ref_spps and (ref_spps[1:4] == ([2, 2, 2] if reset_rest else [3, 5, 7]))
    assert abs(rays - ref_rays) <= 3                                  # (a batched frame reports an equal share of the batch's counts)
    for (a, au), (b, bu) in zip(images, ref_images):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(au, bu)
    assert not np.array_equal(ref_images[1][0], ref_images[2][0])


def _moved(cam, k):
    """the camera of frame k: a small orbit step around the view (what a host's fly-through does every frame)"""
    c = abi.Camera()
    a = 0.03 * k
    d = np.asarray(cam.dir[:], np.float32)
    up = np.asarray(cam.up[:], np.float32)
    right = np.cross(d, up).astype(np.float32)
    nd = (np.cos(a) * d + np.sin(a) * right).astype(np.float32)
    nd /= np.float32(np.sqrt(np.dot(nd, nd)))
    c.pos[:] = [float(cam.pos[0] + 0.05 * k), float(cam.pos[1] + 0.02 * k), float(cam.pos[2] - 0.04 * k)]
    c.dir[:] = [float(x) for x in nd]
    c.up[:] = [float(x) for x in up]
    c.fovy = float(cam.fovy + 0.5 * k)
    return c


@pytest.mark.parametrize("scene_name,reset_rest", [("grid", True), ("grid", False), ("textured_test", True)])
def test_batched_frames_with_a_camera_per_frame_equal_frames_rendered_one_by_one(scene_name, reset_rest):
    """rptr_hip_render_batch_cameras_async: the reference's loop may move the camera every frame (app.cpp:350-469); a launch sequence
    of 4 frames x 2 spp with four different views gives, frame by frame, the image -- and for the last frame the AOV images, whose
    motion vectors are against the third frame's view -- of the same frames submitted alone. Textured scene: the footprint of the
    camera rays follows the frame's camera too."""
    s = scenes.grid(120, 60, with_emitters=True) if scene_name == "grid" else scenes.textured_test()
    W, H, spp, n = 168, 96, 2, 4
    cams = [_moved(s.camera_params(), k) for k in range(n + 1)]

    def run(batched):
        r = backend.RenderHip(frames_in_flight=2)
        r.initialize(W, H)
        r.set_scene(s)
        out = []

        def collect(t):
            st = r.wait(t)
            img = np.zeros((H, W, 4), np.float32)
            assert r.readback_framebuffer(img) == W * H * 4
            out.append((img, st.spp))
        collect(r.render_async(backend.RenderConfiguration(cams[n], active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1))
        if batched:
            cfg0 = backend.RenderConfiguration(cams[0], active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
            for t in r.render_batch_cameras_async(cfg0, cams[:n], spp=spp, reset_rest=reset_rest):
                collect(t)
        else:
            for k in range(n):
                collect(r.render_async(backend.RenderConfiguration(cams[k], active_variant=abi.VARIANT_GLTF, reset_accumulation=(k == 0 or reset_rest)), spp=spp))
        aovs = []
        for i in range(3):
            buf = np.zeros((H, W, 4), np.uint16)
            assert r.readback_aov(i, buf) == W * H * 4
            aovs.append(buf)
        collect(r.render_async(backend.RenderConfiguration(cams[n - 1], active_variant=abi.VARIANT_GLTF, reset_accumulation=False), spp=1))
        r.close()
        return out, aovs

    ref, ref_aovs = run(False)
    got, aovs = run(True)
    assert [g[1] for g in got] == [x[1] for x in ref]
    for k, (a, b) in enumerate(zip(got, ref)):
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), k
    for a, b in zip(aovs, ref_aovs):
        assert np.array_equal(a, b)
    assert not np.array_equal(ref[1][0], ref[2][0])
    if reset_rest:                                                    # different views, not the same image four times
        assert np.abs(ref[1][0][..., :3] - ref[4][0][..., :3]).mean() > 1e-3


def test_batch_limits_and_gather_of_batched_frames():
    s = scenes.cornell32()
    W, H = 96, 64
    cam = s.camera_params()
    cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    one = backend.RenderHip(frames_in_flight=1)
    one.initialize(W, H)
    one.set_scene(s)
    with pytest.raises(backend.BackendError):
        one.render_batch_async(cfg, spp=1, n_frames=2)               # a batch needs per-frame images: frames_in_flight >= 2
    want = []
    for k in range(4):
        one.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=2)
        img = np.zeros((H, W, 4), np.float32)
        one.readback_framebuffer(img)
        want.append(img)
    one.close()
    rs = [backend.RenderHip(rank=k, world_size=2, stripe_rows=8, frames_in_flight=2) for k in range(2)]
    for r in rs:
        r.initialize(W, H)
        r.set_scene(s)
    with pytest.raises(backend.BackendError):
        rs[0].render_batch_async(cfg, spp=1, n_frames=9)             # more than option "max_batch_frames" (8)
    with pytest.raises(backend.BackendError):
        rs[0].render_batch_async(cfg, spp=8, n_frames=3)             # 24 sample slots do not fit
    backend.RenderHip.comm_init_all(rs)
    tickets = [r.render_batch_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=2, n_frames=4) for r in rs]
    for k in range(4):                                               # frame by frame: wait on every rank, gather, compare with one rank
        for r, t in zip(rs, tickets):
            r.wait(t[k])
        backend.RenderHip.gather_all(rs)
        got = np.zeros((H, W, 4), np.float32)
        assert rs[0].readback_gathered(got) == W * H * 4
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32))
    for r in rs:
        r.close()


@pytest.mark.parametrize("transport", ["copy", "peer", "rccl-self"])
def test_one_gather_for_the_frames_of_a_launch_sequence(transport, monkeypatch):
    """rptr_hip_gather_batch / _gather_all_batch: the frames of a launch sequence travel in ONE collective (one transfer per rank, one
    assembly pass) after the sequence's last ticket was waited for -- image k of the gather equals frame k rendered by one rank, bit
    for bit; a gather of the last two frames only; the limits (more frames than the sequence has, than a sequence can have, a batch
    before the last frame was waited for). Transports: the one-process group's device copies and peer writes (three ranks on this
    device), and RCCL through the self-send of world 1."""
    s = scenes.grid(80, 40, with_emitters=True)
    W, H, spp = 96, 64, 2
    cam = s.camera_params()

    def cfg():
        return backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    one = backend.RenderHip(frames_in_flight=1)
    one.initialize(W, H)
    one.set_scene(s)
    want = []
    for k in range(8):
        one.render(cfg(), spp=spp)
        img = np.zeros((H, W, 4), np.float32)
        one.readback_framebuffer(img)
        want.append(img)
    one.close()
    if transport == "rccl-self":
        monkeypatch.setenv("RPTR_COMM_SELF", "1")
        rs = [backend.RenderHip(frames_in_flight=2)]
    else:
        monkeypatch.setenv("RPTR_COMM_TRANSPORT", transport)
        rs = [backend.RenderHip(rank=k, world_size=3, stripe_rows=8, frames_in_flight=2) for k in range(3)]
    for r in rs:
        r.initialize(W, H)
        r.set_scene(s)
    if transport == "rccl-self":
        rs[0].comm_init_rank(backend.RenderHip.comm_unique_id())

        def gather(n):
            rs[0].gather(n)
    else:
        backend.RenderHip.comm_init_all(rs)

        def gather(n):
            backend.RenderHip.gather_all(rs, n)
    got = np.zeros((H, W, 4), np.float32)
    # sequence 1: four frames, one gather
    tickets = [r.render_batch_async(cfg(), spp=spp, n_frames=4) for r in rs]
    for r, t in zip(rs, tickets):
        r.wait(t[1])
    with pytest.raises(backend.BackendError):
        gather(4)                                  # frame 1 of the sequence was waited for last: frames 0..1 are all a gather can take
    for r, t in zip(rs, tickets):
        r.wait(t[3])
    with pytest.raises(backend.BackendError):
        gather(5)                                  # more than a launch sequence holds
    gather(4)
    for k in range(4):
        assert rs[0].readback_gathered(got, k) == W * H * 4
        assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), k
    rs[0].readback_gathered(got)                   # the default: the last frame of the gather
    assert np.array_equal(got.view(np.uint32), want[3].view(np.uint32))
    with pytest.raises(backend.BackendError):
        rs[0].readback_gathered(got, 4)
    # sequence 2 (the other frame context, the other slot on rank 0): the last two of its four frames only
    tickets = [r.render_batch_async(cfg(), spp=spp, n_frames=4) for r in rs]
    for r, t in zip(rs, tickets):
        r.wait(t[3])
    gather(2)
    for k in range(2):
        rs[0].readback_gathered(got, k)
        assert np.array_equal(got.view(np.uint32), want[6 + k].view(np.uint32)), k
    with pytest.raises(backend.BackendError):
        rs[0].readback_gathered(got, 2)
    assert rs[0].comm_stats()[0] == 2
    for r in rs:
        r.close()


@pytest.mark.parametrize("transport", ["copy", "peer"])
def test_gather_all_when_some_ranks_own_no_rows(transport, monkeypatch):
    """a frame of 20 rows in stripes of 8 has three stripes: of five ranks, two own nothing. They still take part in every gather (zero
    rows), render (nothing) and report frames; the assembled frame is the one-rank frame"""
    monkeypatch.setenv("RPTR_COMM_TRANSPORT", transport)
    s = scenes.cornell32()
    W, H, spp, world = 64, 20, 2, 5
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    ref = backend.RenderHip()
    ref.initialize(W, H)
    ref.set_scene(s)
    ref.render(cfg, spp=spp)
    want = np.zeros((H, W, 4), np.float32)
    ref.readback_framebuffer(want)
    ref.close()
    rs = [backend.RenderHip(rank=k, world_size=world, stripe_rows=8, frames_in_flight=2) for k in range(world)]
    for r in rs:
        r.initialize(W, H)
        r.set_scene(s)
    assert [r.local_pixel_count() for r in rs] == [8 * W, 8 * W, 4 * W, 0, 0]
    backend.RenderHip.comm_init_all(rs)
    for _ in range(2):
        tickets = [r.render_async(cfg, spp=spp) for r in rs]
        for r, t in zip(rs, tickets):
            r.wait(t)
        backend.RenderHip.gather_all(rs)
    got = np.zeros((H, W, 4), np.float32)
    assert rs[0].readback_gathered(got) == W * H * 4
    # (the second frame: reset again -> frame_offset moved on by spp)
    ref = backend.RenderHip()
    ref.initialize(W, H)
    ref.set_scene(s)
    for _ in range(2):
        ref.render(cfg, spp=spp)
    ref.readback_framebuffer(want)
    ref.close()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for r in rs:
        r.close()


# ---------------------------------------------------------------- C2 as bench.py runs it
@pytest.mark.parametrize("rng_variant", [abi.RNG_VARIANT_UNIFORM, abi.RNG_VARIANT_Z_SBL])
def test_c2_full_size_on_the_benchmarked_schedule(rng_variant):
    """BASELINE.json configs[1] exactly as the bench line times it: 1 M triangles, 1920x1080, 4 spp, Lambert, 11 frame contexts, four frames
    per launch sequence, every frame restarting the accumulation (frame k is seeded with frame_offset = 4 k). Frames 0, 5 and 11 of a
    12-frame run against the oracle on bands of rows (RMSE < 1e-3, coverage identical), the ray counts of every frame against each other,
    and the whole run against the same frames rendered one at a time on a fresh handle, bit for bit. Also with the Z-Sobol point set."""
    from realtimepathtracingresearchframework_amd import pointsets
    s = scenes.grid_1m()
    W, H, spp, frames = 1920, 1080, 4, 12
    cam = s.camera_params()
    table = pointsets.default_table(rng_variant) if rng_variant != abi.RNG_VARIANT_UNIFORM else None

    def run(fif, batch):
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(s)
        if rng_variant != abi.RNG_VARIANT_UNIFORM:
            r.set_rng_variant(rng_variant, table)
        images, stats, queue, left = [], [], [], frames

        def collect(tickets):
            for t in tickets:
                stats.append(r.wait(t))
                img = np.zeros((H, W, 4), np.float32)
                r.readback_framebuffer(img)
                images.append(img)
        while left > 0:
            cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
            n = min(batch, left)
            queue.append(r.render_batch_async(cfg, spp=spp, n_frames=n, reset_rest=True) if n > 1 else [r.render_async(cfg, spp=spp)])
            left -= n
            if len(queue) >= fif:
                collect(queue.pop(0))
        while queue:
            collect(queue.pop(0))
        r.close()
        return images, stats
    fast, fast_stats = run(11, 4)
    slow, slow_stats = run(1, 1)
    for k in range(frames):
        assert np.array_equal(fast[k].view(np.uint32), slow[k].view(np.uint32)), k
        assert fast_stats[k].spp == spp and W * H * spp <= fast_stats[k].raw.rays_closest <= 9 * W * H * spp
    # the frames of a launch sequence share its counters (every frame reports a quarter of the sequence's rays): sums per sequence
    for g in range(0, frames, 4):
        for field in ("rays_closest", "rays_shadow", "hits_shaded"):
            together = sum(getattr(fast_stats[k].raw, field) for k in range(g, g + 4))
            apart = sum(getattr(slow_stats[k].raw, field) for k in range(g, g + 4))
            assert abs(together - apart) < 4, (g, field, together, apart)
    assert not np.array_equal(fast[0], fast[1])                      # another frame_offset, other samples
    osc = O.OracleScene(s)
    if rng_variant != abi.RNG_VARIANT_UNIFORM:
        osc.set_rng_variant(rng_variant, table)
    osc.build_bvh()
    # the ray counts of a frame rendered on its own are the oracle's, to the ray (SURVEY 8d counting rule)
    _, ost = osc.render(W, H, spp, variant=abi.VARIANT_SIMPLE, frame_offset=0)
    assert (ost.rays_closest, ost.rays_shadow, ost.hits_shaded) == (slow_stats[0].raw.rays_closest, slow_stats[0].raw.rays_shadow, slow_stats[0].raw.hits_shaded)
    for k, rows in ((0, (556, 560)), (5, (700, 704)), (11, (1000, 1004))):
        ref, _ = osc.render(W, H, spp, variant=abi.VARIANT_SIMPLE, rows=rows, frame_offset=spp * k)
        rmse, same, maxabs = image_error(fast[k][rows[0]:rows[1]], ref[rows[0]:rows[1]])
        assert same and rmse < RMSE_TOL, (k, rows, rmse, maxabs)


# ---------------------------------------------------------------- the rest of the adapter surface
def test_ray_queries_on_device_buffers_equal_the_host_array_queries():
    """RenderBackend::enable_ray_queries / render_ray_queries (vulkan/render_vulkan.cpp:430-455,1867-1876): queries and results in device
    buffers the backend owns, traced asynchronously; and rptr_hip_trace_device over a caller's buffers on a caller's stream"""
    import torch
    from common import random_queries
    s = scenes.two_level_test()
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    rng = np.random.default_rng(7)
    n = 20000
    q = random_queries(rng, n, -6, 6)
    q[::7, 3] = np.int32(-1).view(np.float32)                       # mode_or_data < 0: the result slot is left alone
    ref = r.render_ray_queries(q, np.full((n, 4), 7.0, np.float32))
    dq, dr = r.enable_ray_queries_device(n)
    tq = torch.from_numpy(q).cuda()
    tr = torch.full((n, 4), 7.0, dtype=torch.float32, device="cuda")
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    torch.cuda.synchronize()
    assert hip.hipMemcpy(ctypes.c_void_p(dq), ctypes.c_void_p(tq.data_ptr()), ctypes.c_size_t(n * 32), 3) == 0      # device to device
    assert hip.hipMemcpy(ctypes.c_void_p(dr), ctypes.c_void_p(tr.data_ptr()), ctypes.c_size_t(n * 16), 3) == 0
    # (a device-to-device hipMemcpy may return before the copy has run, and the backend's stream does not synchronise with the null stream:
    # without this the 7.0 fill raced the query kernel -- seen once, on the build whose kernels got faster)
    assert hip.hipDeviceSynchronize() == 0
    r.render_ray_queries_device(n)
    r.render_ray_queries(q[:1])                                      # (a synchronous call on the backend's stream: the queries above are done)
    out = torch.empty_like(tr)
    assert hip.hipMemcpy(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(dr), ctypes.c_size_t(n * 16), 3) == 0
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    with pytest.raises(backend.BackendError):
        r.render_ray_queries_device(n + 1)                           # beyond the budget
    # a caller's buffers on a caller's stream
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        r.trace_device(tq.data_ptr(), n, tr.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    assert np.array_equal(tr.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    r.close()


def test_light_sampling_variant_none_equals_a_scene_without_a_light_array():
    """RenderBackendOptions::light_sampling_variant = NONE (rendering/mc/nee.glsl:12-14): no next-event estimation towards the emissive
    triangles; emitters a path hits still shine (at full weight). Bit-identical to the same scene handed over without its light array
    (sun_radiance.w = 1) wherever that image is defined, different from binned RIS, and RIS comes back unchanged"""
    import copy
    s = scenes.grid(120, 60, with_emitters=True)
    W, H, spp = 160, 96, 4
    bare = copy.copy(s)
    bare.lights = np.zeros((0, 4, 3), dtype=np.float32)
    def fresh(variant):                                              # (a fresh handle each: frame_offset advances with every reset)
        r = backend.RenderHip()
        r.initialize(W, H)
        r.set_scene(s)
        r.set_light_sampling_variant(0)                              # toggled: RIS must come back unchanged
        r.set_light_sampling_variant(variant)
        img, st, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, renderer=r)
        r.close()
        return img, st
    ris, _ = fresh(1)
    none, st = fresh(0)
    ris2, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    ref, st_ref, _ = gpu_render(bare, W, H, spp, abi.VARIANT_GLTF)
    assert np.isfinite(none).all()
    # (paths that HIT an emitter: without a light array the MIS weight of that hit is 0 / 0; with the variant switched off it is 1)
    ok = np.isfinite(ref).all(axis=2)
    assert ok.mean() > 0.9 and np.array_equal(none[ok].view(np.uint32), ref[ok].view(np.uint32)) and int(st.raw.rays_shadow) == int(st_ref.raw.rays_shadow)
    assert np.array_equal(ris.view(np.uint32), ris2.view(np.uint32)) and not np.array_equal(ris, none)
