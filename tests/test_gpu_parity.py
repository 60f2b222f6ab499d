"""GPU parity tests: the HIP path, called through the C ABI (librptr_hip.so), against
the CPU oracle on the same seeded inputs; plus size-independent properties at the
benchmark's full size. Tolerances:
  * ray queries (integer/byte/index work + IEEE +,-,*,/ only): bit exact;
  * images: per-pixel RMSE over RGB < 1e-3 (north_star) -- the residual comes from
    libm vs device sin/cos/exp/acos in the last ulps;
  * ray / node / triangle counts: equal up to branch flips caused by those ulps
    (relative 1e-3), exactly equal when traversing identical rays.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
from common import RMSE_TOL, assert_ray_visit_parity, gpu_render, image_error, random_queries
from realtimepathtracingresearchframework_amd import abi, backend, scenes
from realtimepathtracingresearchframework_amd import distributed as D

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small_scenes():
    return {"cornell": scenes.cornell32(), "two_level": scenes.two_level_test(), "grid": scenes.grid(120, 60),
            "grid_lights": scenes.grid(120, 60, with_emitters=True)}


def test_extension_is_loaded_and_names_itself(hip_lib):
    r = backend.RenderHip()
    assert "HIP" in r.name() and r.variant_names() == abi.VARIANT_NAMES
    r.close()


# ---------------------------------------------------------------- RQ_CLOSEST: bit exact
@pytest.mark.parametrize("name", ["cornell", "two_level", "grid"])
def test_trace_closest_bit_exact_vs_brute_force(small_scenes, name):
    s = small_scenes[name]
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    rng = np.random.default_rng(1)
    lo, hi = (-60, 60) if name.startswith("grid") else (-6, 6)
    q = random_queries(rng, 20000, lo, hi)
    if name.startswith("grid"):
        q[:, 1] = np.abs(q[:, 1]) * 0.2 + 1.0  # start above the height field
        q[:, 5] = -np.abs(q[:, 5])
    q[:100, 3] = np.array([-1], np.int32).view(np.float32)[0]  # mode < 0: slot untouched
    res = np.full((len(q), 4), 3.0, np.float32)
    r.render_ray_queries(q, res)
    osc = O.OracleScene(s)
    ref = np.full((len(q), 4), 3.0, np.float32)
    osc.trace(q, bvh_mode=O.BVH_OWN if name == "grid" else O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32))
    assert (res[100:, 0] >= 0).sum() > 500 and (res[:100] == 3.0).all()
    r.close()


def test_exported_bvh_gives_identical_visit_counts(small_scenes):
    """the roofline's algorithmic bytes are *counted*: GPU counters == oracle walking the same tree."""
    s = small_scenes["grid"]
    W, H = 160, 90
    img, st, r = gpu_render(s, W, H, 1, abi.VARIANT_SIMPLE, count=True, keep=True)
    osc = O.OracleScene(s)
    osc.import_bvh(*r.export_bvh())
    ref, ost = osc.render(W, H, 1, variant=abi.VARIANT_SIMPLE, bvh_mode=O.BVH_IMPORTED, count=True)
    assert st.raw.rays_closest == ost.rays_closest and st.raw.rays_shadow == ost.rays_shadow
    assert st.raw.nodes_visited == ost.nodes_closest + ost.nodes_shadow
    assert st.raw.tris_tested == ost.tris_closest + ost.tris_shadow
    assert st.raw.hits_shaded == ost.hits_shaded
    r.close()


@pytest.mark.parametrize("name,variant", [("cornell", abi.VARIANT_GLTF), ("two_level", abi.VARIANT_GLTF), ("grid_lights", abi.VARIANT_SIMPLE)])
def test_every_ray_of_a_frame_walks_the_tree_like_the_oracle(small_scenes, name, variant):
    s = small_scenes[name]
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    assert_ray_visit_parity(r, O.OracleScene(s), 96, 64, 2, variant)
    r.close()


# ---------------------------------------------------------------- images vs oracle
@pytest.mark.parametrize("name,variant,W,H,spp", [
    ("cornell", abi.VARIANT_GLTF, 128, 128, 3),
    ("cornell", abi.VARIANT_SIMPLE, 96, 64, 2),
    ("two_level", abi.VARIANT_GLTF, 160, 120, 2),
    ("grid", abi.VARIANT_SIMPLE, 240, 136, 2),
    ("grid", abi.VARIANT_GLTF, 240, 136, 2),
    ("grid_lights", abi.VARIANT_GLTF, 240, 136, 2),
])
def test_image_parity_vs_oracle(small_scenes, name, variant, W, H, spp):
    s = small_scenes[name]
    img, st, _ = gpu_render(s, W, H, spp, variant)
    ref, ost = O.OracleScene(s).render(W, H, spp, variant=variant)
    rmse, same_nan_mask, maxabs = image_error(img, ref)
    assert same_nan_mask
    assert rmse < RMSE_TOL, (rmse, maxabs)
    assert np.array_equal(np.nan_to_num(img[..., 3]), np.nan_to_num(ref[..., 3]))  # alpha = "hit something"
    assert abs(int(st.raw.rays_closest) - int(ost.rays_closest)) <= max(4, 1e-3 * ost.rays_closest)
    assert abs(int(st.raw.rays_shadow) - int(ost.rays_shadow)) <= max(4, 1e-3 * ost.rays_shadow)
    assert st.spp == spp


def test_progressive_accumulation_matches_one_shot(small_scenes):
    """N frames of 1 spp (reference: batch_spp = 1 + running mean) == one call with N spp, bit for bit."""
    s = small_scenes["cornell"]
    W = H = 64
    one, _, _ = gpu_render(s, W, H, 4, abi.VARIANT_GLTF)
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    for k in range(4):
        cfg = backend.RenderConfiguration(s.camera_params(), reset_accumulation=(k == 0))
        st = r.render(cfg, spp=1)
    assert st.spp == 4
    img = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img)
    assert np.array_equal(img, one)
    # reset_accumulation moves the seed: frame_offset += frame_id (render_vulkan.cpp:1937-1941)
    cfg = backend.RenderConfiguration(s.camera_params(), reset_accumulation=True)
    r.render(cfg, spp=4)
    img2 = np.zeros_like(img)
    r.readback_framebuffer(img2)
    assert not np.array_equal(img2, one)
    ref, _ = O.OracleScene(s).render(W, H, 4, frame_offset=4)
    assert image_error(img2, ref)[0] < RMSE_TOL
    r.close()


def test_small_batches_equal_large_batches(small_scenes, monkeypatch):
    s = small_scenes["grid"]
    a, _, _ = gpu_render(s, 128, 72, 5, abi.VARIANT_SIMPLE)
    monkeypatch.setenv("RPTR_MAX_BATCH_SPP", "2")
    b, _, _ = gpu_render(s, 128, 72, 5, abi.VARIANT_SIMPLE)
    assert np.array_equal(a, b)


def test_traversal_thresholds_follow_the_tree_and_change_no_bit(monkeypatch):
    """the scheduling thresholds of the traversal (when a wave refills its idle lanes, when it leaves a node phase) are chosen per scene from
    the surface-area cost of its tree (rptr_hip_traversal_preset): the dense forest gets 16 / 32, the height field the defaults -- and whatever
    they are, image, ray counts and counted node / triangle visits stay the same bit for bit (they decide when a lane steps, not what it finds)"""
    forest = scenes.forest(n_meshes=4, tris_per_tree=3000, n_instances=120, name="small-forest")
    field = scenes.grid(160, 80)
    for flatten in ("1", "0"):
        monkeypatch.setenv("RPTR_FLATTEN", flatten)
        r = backend.RenderHip()
        r.initialize(64, 48)
        r.set_scene(forest)
        cost, nm, rm = r.traversal_preset()
        r.close()
        assert cost >= 24.0 and (nm, rm) == (16, 32), (flatten, cost, nm, rm)
    monkeypatch.setenv("RPTR_FLATTEN", "0")
    r = backend.RenderHip()
    r.initialize(64, 48)
    r.set_scene(field)
    cost, nm, rm = r.traversal_preset()
    r.close()
    assert cost < 24.0 and (nm, rm) == (0, 0), (cost, nm, rm)
    monkeypatch.setenv("RPTR_FLATTEN", "1")
    out = []
    for preset in (None, "0,0", "40,8", "1,64"):
        if preset is None:
            monkeypatch.delenv("RPTR_TRAVERSE_PRESET", raising=False)
        else:
            monkeypatch.setenv("RPTR_TRAVERSE_PRESET", preset)
        img, st, _ = gpu_render(forest, 200, 120, 2, abi.VARIANT_GLTF, count=True)
        out.append((img, int(st.raw.rays_closest), int(st.raw.rays_shadow), int(st.raw.nodes_visited), int(st.raw.tris_tested)))
    for other in out[1:]:
        assert np.array_equal(out[0][0].view(np.uint32), other[0].view(np.uint32)) and out[0][1:] == other[1:]


def test_lds_staged_top_of_the_tree_changes_no_bit(monkeypatch):
    """north_star's "LDS-staged node tiles" (RPTR_LDS_TOP=1: the first 64 nodes of the breadth-first node array are copied into LDS when a
    traversal block starts and read from there; a separate instantiation of the closest-hit / shadow kernels for plain single-instance
    scenes; off by default -- it does not pay, profiles/r03_notes.md section 8): same image and ray counts bit for bit, one frame at a time
    and with frames in flight"""
    s = scenes.grid(200, 100, with_emitters=True)
    W, H, spp = 256, 144, 3
    out = []
    for top in ("0", "1"):
        monkeypatch.setenv("RPTR_LDS_TOP", top)
        img, st, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
        r = backend.RenderHip(frames_in_flight=3)
        r.initialize(W, H)
        r.set_scene(s)
        cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
        tickets = [r.render_async(cfg, spp=spp) for _ in range(3)]
        for t in tickets:
            r.wait(t)
        img2 = np.zeros((H, W, 4), np.float32)
        r.readback_framebuffer(img2)
        r.close()
        out.append((img, img2, int(st.raw.rays_closest), int(st.raw.rays_shadow)))
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)) and np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
    assert out[0][2:] == out[1][2:] and out[0][3] > 0


def test_regrouping_by_material_does_not_change_the_image(small_scenes, monkeypatch):
    """north_star's regrouping of rays by material, fused into the shade kernel's LDS compaction (RPTR_REGROUP=1; off by default: it costs
    6 % of the shade time on C3 with 48 textured materials and gains nothing, profiles/r03_notes.md): a path's result does not depend on
    its position in a chunk, so image and ray counts are those of the plain schedule, bit for bit -- one frame at a time and pipelined"""
    s = small_scenes["grid_lights"]
    monkeypatch.setenv("RPTR_REGROUP", "1")
    a, sa, _ = gpu_render(s, 320, 180, 2, abi.VARIANT_GLTF)   # chunks of 1024 paths with > 64 hits: the ordering really runs
    ta, _, _ = gpu_render(scenes.textured_test(), 160, 96, 3, abi.VARIANT_GLTF)
    monkeypatch.setenv("RPTR_REGROUP", "0")
    b, sb, _ = gpu_render(s, 320, 180, 2, abi.VARIANT_GLTF)
    tb, _, _ = gpu_render(scenes.textured_test(), 160, 96, 3, abi.VARIANT_GLTF)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.raw.rays_shadow == sb.raw.rays_shadow and sa.raw.rays_closest == sb.raw.rays_closest
    assert np.array_equal(ta.view(np.uint32), tb.view(np.uint32))


# ---------------------------------------------------------------- tiles: N ranks == 1 rank, bit identical
@pytest.mark.parametrize("world,stripe", [(2, 32), (3, 8), (8, 16)])
def test_tile_split_is_bit_identical_to_single_gpu(small_scenes, world, stripe):
    s = small_scenes["two_level"]
    W, H = 136, 100
    full, st_full, _ = gpu_render(s, W, H, 2, abi.VARIANT_GLTF)
    frame = np.zeros_like(full)
    rays = 0
    for rank in range(world):
        part = np.full_like(full, -7.0)
        r = backend.RenderHip(rank=rank, world_size=world, stripe_rows=stripe)
        r.initialize(W, H)
        r.set_scene(s)
        st = r.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=True), spp=2)
        r.readback_framebuffer(part)
        rows = r.tile_rows()
        assert rows == D.tile_rows(H, stripe, rank, world)
        assert r.local_pixel_count() == sum(c for _, c in rows) * W
        mask = np.zeros(H, bool)
        for first, cnt in rows:
            mask[first:first + cnt] = True
            frame[first:first + cnt] = part[first:first + cnt]
        assert (part[~mask] == -7.0).all()  # rows of other ranks untouched
        rays += st.raw.rays_closest
        r.close()
    assert np.array_equal(np.nan_to_num(frame, nan=-1), np.nan_to_num(full, nan=-1))
    assert rays == st_full.raw.rays_closest


def test_copy_tile_to_device_matches_readback(small_scenes):
    import torch
    s = small_scenes["cornell"]
    W, H = 64, 48
    r = backend.RenderHip(rank=1, world_size=2, stripe_rows=8, stream=torch.cuda.current_stream().cuda_stream)
    r.initialize(W, H)
    r.set_scene(s)
    r.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=True), spp=1)
    n = r.local_pixel_count()
    t = torch.zeros((n, 4), dtype=torch.float32, device="cuda")
    r.copy_tile_to_device(t.data_ptr(), t.numel() * 4)
    torch.cuda.synchronize()
    img = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img)
    packed = np.concatenate([img[f:f + c].reshape(-1, 4) for f, c in r.tile_rows()])
    assert np.array_equal(t.cpu().numpy(), packed)
    r.close()


# ---------------------------------------------------------------- error behaviour of the boundary
def test_error_convention(small_scenes):
    r = backend.RenderHip()
    with pytest.raises(backend.BackendError):
        r.render(backend.RenderConfiguration(small_scenes["cornell"].camera_params()), spp=1)  # before set_scene
    r.initialize(32, 32)
    bad = scenes.cornell32()
    bad.materials[0].normal_map = 3                          # not a texture of the scene
    with pytest.raises(backend.BackendError) as e:
        r.set_scene(bad)
    assert e.value.code == abi.RPTR_E_INVALID
    r.set_scene(small_scenes["cornell"])
    assert r.readback_framebuffer(np.zeros(10, np.float32)) == 0  # too small -> 0 (render_vulkan.cpp:2262-2263)
    assert r.configure_for(None, 7) is False
    r.close()


# ---------------------------------------------------------------- full-size properties (BASELINE.json configs[1])
@pytest.fixture(scope="module")
def grid_1m():
    return scenes.grid_1m()


def test_full_size_determinism_and_counts(grid_1m):
    W, H = 1920, 1080
    a, sa, r = gpu_render(grid_1m, W, H, 1, abi.VARIANT_SIMPLE, keep=True)
    r.close()
    # same seed again needs a fresh handle (a reset on the same handle advances frame_offset)
    b, sb, _ = gpu_render(grid_1m, W, H, 1, abi.VARIANT_SIMPLE)
    assert np.array_equal(a, b) and sa.raw.rays_closest == sb.raw.rays_closest
    assert np.isfinite(a).all()
    assert sa.raw.rays_closest >= W * H                       # one primary ray per pixel sample
    assert sa.raw.rays_closest <= W * H * 9 and sa.raw.hits_shaded < sa.raw.rays_closest
    # a band of rows against the oracle at the full frame size (same pixels, same seeds)
    rows = (560, 568)
    osc = O.OracleScene(grid_1m)
    ref, _ = osc.render(W, H, 1, variant=abi.VARIANT_SIMPLE, rows=rows)
    rmse, same, _ = image_error(a[rows[0]:rows[1]], ref[rows[0]:rows[1]])
    assert same and rmse < RMSE_TOL


def test_full_size_queries_sorted_and_consistent(grid_1m):
    """property: for camera-like rays, hits reported by RQ_CLOSEST are reproducible and barycentrics are valid."""
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(grid_1m)
    rng = np.random.default_rng(3)
    n = 1 << 20
    q = random_queries(rng, n, -50, 50)
    q[:, 1] = 10 + np.abs(q[:, 1]) * 0.1
    q[:, 5] = -np.abs(q[:, 5]) - 0.05
    res = r.render_ray_queries(q)
    res2 = r.render_ray_queries(q)
    assert np.array_equal(res.view(np.uint32), res2.view(np.uint32))
    hit = res[:, 0] >= 0
    assert hit.mean() > 0.3
    assert (res[hit, 0] + res[hit, 1] <= 1.0 + 1e-6).all()
    prim = res[hit, 3].view(np.int32)
    assert prim.min() >= 0 and prim.max() < 1000000
    r.close()


# ---------------------------------------------------------------- instanced forest
def test_small_forest_trace_and_image_parity():
    """two-level hierarchy with overlapping instances of shared meshes: 4 meshes x 400 triangles, 36 instances."""
    s = scenes.forest(n_meshes=4, tris_per_tree=400, n_instances=36, name="forest-small")
    W, H, spp = 160, 90, 2
    img, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, count=True, keep=True)
    osc = O.OracleScene(s)
    q = random_queries(np.random.default_rng(8), 20000, -7, 7)
    q[:, 1] = np.abs(q[:, 1]) + 0.2
    res = r.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).mean() > 0.3
    ref_img, _ = osc.render(W, H, spp, variant=abi.VARIANT_GLTF)
    rmse, same, _ = image_error(img, ref_img)
    assert same and rmse < RMSE_TOL
    ost = assert_ray_visit_parity(r, osc, W, H, spp, abi.VARIANT_GLTF)   # every ray, both query kinds, exact
    # whole-render counters: the GPU's own rays differ from the oracle's by an ulp here and there (libm), hence 1e-4
    assert st.raw.rays_closest == ost.rays_closest and st.raw.rays_shadow == ost.rays_shadow
    assert abs(int(st.raw.nodes_visited) - (ost.nodes_closest + ost.nodes_shadow)) <= 1e-4 * st.raw.nodes_visited
    assert abs(int(st.raw.tris_tested) - (ost.tris_closest + ost.tris_shadow)) <= 1e-4 * st.raw.tris_tested
    r.close()


def test_full_forest_10m_instanced_triangles():
    """C4 at full size: 10 meshes x 10k triangles, 1000 instances + ground = 10 000 002 instanced triangles. Properties:
    deterministic, every hit names a valid (geometry, primitive), and a band of rows equals the oracle."""
    s = scenes.forest()
    assert s.num_instanced_tris() == 10_000_002
    W, H = 1920, 1080
    a, sa, r = gpu_render(s, W, H, 1, abi.VARIANT_GLTF, keep=True)
    q = random_queries(np.random.default_rng(5), 1 << 18, -30, 30)
    q[:, 1] = np.abs(q[:, 1]) * 0.3 + 0.5
    res = r.render_ray_queries(q)
    hit = res[:, 0] >= 0
    assert hit.mean() > 0.5
    geom, prim = res[hit, 2].view(np.int32), res[hit, 3].view(np.int32)   # instanced-geometry index: 10 trees + ground
    assert geom.min() >= 0 and geom.max() <= 10 and prim.min() >= 0 and prim.max() < 10000
    assert (res[hit, 0] + res[hit, 1] <= 1.0 + 1e-6).all()
    r.close()
    b, sb, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF)
    assert np.array_equal(a, b) and sa.raw.rays_closest == sb.raw.rays_closest and np.isfinite(a).all()
    rows = (600, 606)
    osc = O.OracleScene(s)
    ref, _ = osc.render(W, H, 1, variant=abi.VARIANT_GLTF, rows=rows)
    rmse, same, _ = image_error(a[rows[0]:rows[1]], ref[rows[0]:rows[1]])
    assert same and rmse < RMSE_TOL


# ---------------------------------------------------------------- frames in flight
def test_frames_in_flight_give_the_same_images_in_the_same_order(small_scenes):
    """3 frame contexts: a progressive sequence (reset, then two accumulating frames, then a reset with another camera)
    queued back to back equals the same sequence rendered one frame at a time, image by image, bit for bit."""
    s = small_scenes["grid_lights"]
    W, H = 160, 90
    cam_a = s.camera_params()
    cam_b = s.camera_params()
    cam_b.pos[0] += 3.0
    seq = [(cam_a, True, 1), (cam_a, False, 2), (cam_a, False, 1), (cam_b, True, 2), (cam_b, False, 1)]

    def run(fif):
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(s)
        images, stats, queue = [], [], []
        def collect():
            st = r.wait(queue.pop(0))
            img = np.zeros((H, W, 4), np.float32)
            assert r.readback_framebuffer(img) == W * H * 4
            images.append(img)
            stats.append((st.raw.rays_closest, st.raw.rays_shadow, st.raw.spp))
        for cam, reset, spp in seq:
            cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=reset)
            queue.append(r.render_async(cfg, spp=spp))
            if len(queue) >= fif:
                collect()
        while queue:
            collect()
        with pytest.raises(backend.BackendError):
            r.wait(12345)                      # not in flight
        r.close()
        return images, stats

    ref_images, ref_stats = run(1)
    images, stats = run(3)
    assert stats == ref_stats
    for a, b in zip(images, ref_images):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert not np.array_equal(ref_images[0], ref_images[1])


def test_more_tickets_than_contexts_is_an_error(small_scenes):
    s = small_scenes["cornell"]
    r = backend.RenderHip(frames_in_flight=2)
    r.initialize(64, 64)
    r.set_scene(s)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    t1 = r.render_async(cfg, spp=1)
    t2 = r.render_async(cfg, spp=1)
    with pytest.raises(backend.BackendError):
        r.render_async(cfg, spp=1)
    r.wait(t1)
    t3 = r.render_async(cfg, spp=1)
    # a synchronous call drains whatever is still queued and then renders
    st = r.render(cfg, spp=1)
    assert st.raw.rays_closest > 0
    with pytest.raises(backend.BackendError):
        r.wait(t2)
    r.close()


# ---------------------------------------------------------------- bench.py's N > 1 control flow on one GPU
def test_bench_two_ranks_on_one_gpu_with_gloo(tmp_path):
    """torch.distributed.run with 2 ranks, both rendering their stripes on cuda:0, tiles gathered over gloo: the same loop
    the driver runs with RCCL on N GPUs (frames in flight, wait, tile copy, gather, max-over-ranks timing, one JSON line)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29517",
           os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--grid", "100x50", "--width", "320", "--height", "180",
           "--dist-backend", "gloo", "--same-device"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["rays_per_step"] > 320 * 180 * 4   # both ranks' rays are summed
    assert "cpu_baseline" not in d


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT torch.distributed.run (the way a driver may call it): bench.py re-executes itself under
    torch.distributed.run, one JSON line comes back"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", "100x50", "--width", "320",
           "--height", "180", "--dist-backend", "gloo", "--same-device"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["value"] > 0 and d["gather"]["mode"].startswith("tile copy")
    rf = d["roofline"]
    assert rf["exclusive_ms_per_step"] <= rf["stage_ms_per_step"]["gpu_total"] + 1e-6
    for k in rf["kernels"].values():
        assert 0.0 <= k["algorithmic_frac"] <= 1.0 and (k["hbm_frac"] is None or 0.0 <= k["hbm_frac"] <= 1.0)


def test_bench_n_ranks_dry_run_probes_rccl_and_falls_back_without_hanging():
    """The exact N > 1 path the driver launches (`bench.py --gpus N`, default --dist-backend nccl), on the one GPU of this box with both
    ranks on cuda:0: the gloo control plane comes up, every rank's probe child tries torch's nccl group and the library's communicator
    under a time-out, RCCL refuses two ranks on one device ("Duplicate GPU"), all ranks agree on the fall-back (tiles over gloo, staged
    through the host), the run completes and the line says what happened. On a real node the same code takes the native path
    (gather.mode "library: grouped ncclSend/ncclRecv ...")."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--grid", "100x50", "--width", "320",
           "--height", "180", "--same-device", "--probe-timeout", "120"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    g = d["gather"]
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["value"] > 0 and d["scaling"] == "strong"
    assert g["probe"]["native"] == 0 and g["mode"].startswith("tile copy") and "gloo" in g["mode"] and g["note"] and "probe" in g["note"].lower()
    assert g["probe"]["seconds"] < 120 and g["gathers"] == 6 and g["bytes_per_step"] > 0
    assert d["config"]["rays_per_step"] > 320 * 180 * 4          # both ranks' rays are summed
    assert d["roofline"]["latency"]["1"]["ms_per_frame"] > 0


@pytest.mark.parametrize("ranks", [2, 3])
def test_ipc_peer_write_gather_between_processes_is_bit_identical_to_one_rank(ranks):
    """One process per rank (torch.distributed.run, both / all three on cuda:0 of this box), the library's peer-write transport for that
    launch mode: rank 0 exports its frame buffers (hipIpcGetMemHandle), the other processes map them and scatter their rows into rank 0's
    frame themselves, ordered by counters in rank 0's flag block -- single frames and launch sequences of four gathered in ONE
    collective; every assembled frame equals the one-rank frame bit for bit (tools/ipc_gather_check.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1", "--master-port",
           str(29720 + ranks), os.path.join(root, "tools", "ipc_gather_check.py")]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "IPC_GATHER_OK 10 frames, gathers 4" in p.stdout, (p.stdout + p.stderr)[-3000:]


def test_bench_n_ranks_with_the_ipc_gather_on_one_device():
    """`bench.py --gpus 2 --gather ipc` with both ranks on cuda:0: the data plane stays on the device (RCCL cannot run there and the default
    run falls back to tiles over gloo), one gather per launch sequence, the line names the transport"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--grid", "100x50", "--width", "320",
           "--height", "180", "--same-device", "--probe-timeout", "120", "--gather", "ipc", "--no-cpu-baseline"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-3000:]
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    g = d["gather"]
    assert d["n_gpus"] == 2 and d["steps"] == 8 and d["value"] > 0
    assert g["transport"] == "ipc" and g["frames_per_gather"] == 4 and g["gathers"] == 2 and g["mode"].startswith("library")
    assert len(g["latency_1_ms_per_rank"]) == 2 and all(x > 0 for x in g["latency_1_ms_per_rank"])     # every rank's own share, one frame at a time


# ---------------------------------------------------------------- fuzz
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_soups_trace_and_image(seed):
    """degenerate triangles, flat meshes, sheared / mirrored instances: ray queries bit-exact vs brute force, every ray of a
    frame walks the exported tree like the oracle, image within tolerance"""
    s = scenes.soup(seed)
    r = backend.RenderHip()
    r.initialize(96, 64)
    r.set_scene(s)
    q = random_queries(np.random.default_rng(seed), 20000, -5, 5)
    res = r.render_ray_queries(q)
    osc = O.OracleScene(s)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).sum() > 500
    assert_ray_visit_parity(r, osc, 96, 64, 1, abi.VARIANT_GLTF)
    img, _, _ = gpu_render(s, 96, 64, 2, abi.VARIANT_GLTF, renderer=r)
    ref_img, _ = osc.render(96, 64, 2, variant=abi.VARIANT_GLTF)
    rmse, same, _ = image_error(img, ref_img)
    assert same and rmse < RMSE_TOL
    r.close()


# ---------------------------------------------------------------- textured materials (a8 / a9)
@pytest.mark.parametrize("variant", [abi.VARIANT_GLTF, abi.VARIANT_SIMPLE])
def test_textured_materials_and_normal_map_image_parity(variant):
    """base colour (sRGB), specular / roughness / metallic channels, a tangent-space normal map and a textured emitter:
    image within tolerance of the oracle, ray counts equal, and the textures do matter"""
    s = scenes.textured_test()
    W, H, spp = 160, 120, 4
    img, st, _ = gpu_render(s, W, H, spp, variant)
    ref, ost = O.OracleScene(s).render(W, H, spp, variant=variant)
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL
    assert st.raw.rays_closest == ost.rays_closest and abs(int(st.raw.rays_shadow) - ost.rays_shadow) <= 1e-3 * ost.rays_shadow
    plain = scenes.textured_test()
    for m in plain.materials:
        m.normal_map = -1
    img2, _, _ = gpu_render(plain, W, H, spp, variant)
    assert image_error(img, img2)[0] > 10 * RMSE_TOL


def test_texture_handles_are_validated():
    s = scenes.textured_test()
    s.materials[0].normal_map = 9
    r = backend.RenderHip()
    r.initialize(32, 32)
    with pytest.raises(backend.BackendError) as e:
        r.set_scene(s)
    assert e.value.code == abi.RPTR_E_INVALID
    s = scenes.textured_test()
    abi.set_float_bits(s.materials[1].base_color, 0, 0x80000000 | 77)
    with pytest.raises(backend.BackendError):
        r.set_scene(s)
    r.close()


# ---------------------------------------------------------------- alpha-tested geometry (a5)
def _oracle_on_device_tree(s, r):
    osc = O.OracleScene(s)
    osc.import_bvh(*r.export_bvh())
    return osc


@pytest.mark.parametrize("variant", [abi.VARIANT_GLTF, abi.VARIANT_SIMPLE])
def test_alpha_tested_geometry_image_parity(variant):
    """cut-outs, fractional alphas (stochastic test, path generator for closest hits, per-candidate generator for shadow
    rays), per-triangle materials, instances: the image equals the oracle's walking the same tree in the same order"""
    s = scenes.alpha_test()
    W, H, spp = 160, 120, 4
    img, st, r = gpu_render(s, W, H, spp, variant, keep=True)
    osc = _oracle_on_device_tree(s, r)
    ref, ost = osc.render(W, H, spp, variant=variant, bvh_mode=O.BVH_IMPORTED)
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL
    assert abs(int(st.raw.rays_closest) - ost.rays_closest) <= 1e-3 * ost.rays_closest
    assert abs(int(st.raw.rays_shadow) - ost.rays_shadow) <= 1e-3 * ost.rays_shadow
    # a second frame accumulates on top (frame_id = 4 enters the seeds of the shadow-ray alpha tests)
    img2, _, _ = gpu_render(s, W, H, spp, variant, reset=False, renderer=r)
    ref2, _ = osc.render(W, H, spp, variant=variant, bvh_mode=O.BVH_IMPORTED, sample_begin=spp, accum=ref.copy())
    rmse2, same2, _ = image_error(img2, ref2)
    assert same2 and rmse2 < RMSE_TOL
    r.close()
    # and the alpha test matters: the same scene with every material opaque
    opaque = scenes.alpha_test()
    for m in opaque.materials:
        m.flags |= abi.BASE_MATERIAL_NOALPHA
    img3, _, _ = gpu_render(opaque, W, H, spp, variant)
    assert image_error(img, img3)[0] > 50 * RMSE_TOL


def test_alpha_tested_geometry_tiles_and_batches_are_bit_identical():
    """the alpha tests only depend on (pixel, sample, frame), not on how the frame is split into stripes or batches"""
    s = scenes.alpha_test()
    W, H, spp = 96, 80, 4
    full, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    parts = np.zeros_like(full)
    for rank in range(3):
        img, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, rank=rank, world=3, stripe_rows=8)
        rows = [y for y in range(H) if (y // 8) % 3 == rank]
        parts[rows] = img[rows]
    assert np.array_equal(full.view(np.uint32), parts.view(np.uint32))
    import os
    os.environ["RPTR_MAX_BATCH_SPP"] = "1"
    try:
        one, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF)
    finally:
        del os.environ["RPTR_MAX_BATCH_SPP"]
    assert np.array_equal(full.view(np.uint32), one.view(np.uint32))


def test_alpha_textures_that_are_opaque_change_nothing():
    """alpha = 1 everywhere: no candidate is rejected, no random number is drawn -> bit-identical to NOALPHA materials"""
    a = scenes.alpha_test()
    b = scenes.alpha_test()
    for sc in (a, b):
        for t in sc.textures:
            t.rgba[..., 3] = 255
    for m in b.materials:
        m.flags |= abi.BASE_MATERIAL_NOALPHA
    ia, _, _ = gpu_render(a, 128, 96, 2, abi.VARIANT_GLTF)
    ib, _, _ = gpu_render(b, 128, 96, 2, abi.VARIANT_GLTF)
    assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32))


# ---------------------------------------------------------------- .vks assets
@pytest.mark.parametrize("version", [3, 4])
def test_vks_scene_renders_like_the_oracle(version):
    """a .vks file + its texture directory (tests/golden/vks, written by vks.py, accepted by the reference's reader) loaded the
    way Scene::load_vkrs does -- every material textured, normal-mapped and, for the RGBA8 colour textures, alpha-tested"""
    import os
    from realtimepathtracingresearchframework_amd import vks
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vks", "alpha_v%d.vks" % version)
    s = vks.read_vks(path)
    src = scenes.alpha_test()
    s.camera, s.config, s.sky_key = src.camera, src.config, src.sky_key
    assert len(s.textures) == 3 * len(s.materials) and all(m.normal_map >= 0 for m in s.materials)
    W, H, spp = 160, 120, 4
    img, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True)
    osc = _oracle_on_device_tree(s, r)
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, bvh_mode=O.BVH_IMPORTED)
    r.close()
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL
    assert abs(int(st.raw.rays_closest) - ost.rays_closest) <= 1e-3 * ost.rays_closest



# ---------------------------------------------------------------- AOV images (RenderGraphic::readback_aov)
def _aov_close(a, b, what):
    """float16 images: the same finite / non-finite pattern and values within two half ulps (the floats behind them differ by
    libm ulps between host and device; an RGBA16F store rounds them)"""
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    fa, fb = np.isfinite(a32), np.isfinite(b32)
    assert (fa == fb).mean() > 0.9995, what
    both = fa & fb
    err = np.abs(a32[both] - b32[both])
    tol = 2.0 ** -9 * np.maximum(np.abs(b32[both]), 2.0 ** -5)
    assert (err <= tol).mean() > 0.999, (what, float(err.max()))


@pytest.mark.parametrize("variant", [abi.VARIANT_GLTF, abi.VARIANT_SIMPLE])
def test_aov_images_match_the_oracle(variant):
    s = scenes.textured_test()
    W, H, spp = 160, 120, 2
    img, st, r = gpu_render(s, W, H, spp, variant, keep=True)
    ref, _, aovs = O.OracleScene(s).render(W, H, spp, variant=variant, aovs=True)
    for k, name in enumerate(("albedo_roughness", "normal_depth", "motion_jitter")):
        got = np.zeros((H, W, 4), np.float16)
        assert r.readback_aov(k, got) == W * H * 4
        _aov_close(got, aovs[k], name)
    # the next frame accumulates; its first sample (index 2) writes the AOVs again, the view did not move
    img2, _, _ = gpu_render(s, W, H, spp, variant, reset=False, renderer=r)
    _, _, aovs2 = O.OracleScene(s).render(W, H, spp, variant=variant, sample_begin=spp, accum=ref.copy(), aovs=True)
    got = np.zeros((H, W, 4), np.float16)
    r.readback_aov(r.AOVNormalDepthIndex, got)
    _aov_close(got, aovs2[1], "normal_depth of the second frame")
    assert r.readback_aov(0, np.zeros(10, np.float16)) == 0          # too small -> 0
    with pytest.raises(backend.BackendError):
        r.readback_aov(5, got)
    r.close()


def test_aov_motion_and_tiles():
    """a moved camera shows up in the motion AOV; a stripe-split frame returns its own rows only"""
    s = scenes.textured_test()
    W, H = 96, 80
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    prev = s.camera_params()
    prev.pos[0] -= 0.25
    r.render(backend.RenderConfiguration(prev, active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1)
    r.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=1)
    got = np.zeros((H, W, 4), np.float16)
    r.readback_aov(r.AOVMotionJitterIndex, got)
    # the reset before the second frame moved frame_offset on by the one sample of the first (render_vulkan.cpp:1937-1941)
    _, _, aovs = O.OracleScene(s).render(W, H, 1, aovs=True, prev_camera=prev, frame_offset=1)
    hit = np.isfinite(aovs[1].astype(np.float32)[..., 3])       # sky pixels project the point (2e32, 2e32, 2e32): inf - inf, not compared
    _aov_close(got[hit], aovs[2][hit], "motion_jitter")
    assert np.abs(got.astype(np.float32)[hit][:, 0]).max() > 0.01
    r.close()
    full = np.zeros((H, W, 4), np.float16)
    parts = np.zeros((H, W, 4), np.float16)
    rf = backend.RenderHip()
    rf.initialize(W, H)
    rf.set_scene(s)
    rf.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=True), spp=2)
    rf.readback_aov(1, full)
    rf.close()
    for rank in range(2):
        rr = backend.RenderHip(rank=rank, world_size=2, stripe_rows=8)
        rr.initialize(W, H)
        rr.set_scene(s)
        rr.render(backend.RenderConfiguration(s.camera_params(), reset_accumulation=True), spp=2)
        rr.readback_aov(1, parts)
        rr.close()
    assert np.array_equal(full.view(np.uint16), parts.view(np.uint16))


def test_empty_scene_is_all_sky():
    """a scene without instances (an empty top level): every query misses, the image is the sky model"""
    s = scenes.cornell32()
    s.instances = []
    s.prepare_lights()
    img, st, r = gpu_render(s, 64, 48, 2, abi.VARIANT_GLTF, keep=True)
    ref, ost = O.OracleScene(s).render(64, 48, 2)
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL and st.raw.rays_closest == 64 * 48 * 2 == ost.rays_closest and st.raw.rays_shadow == 0
    q = np.zeros((5, 8), np.float32)
    q[:, 6], q[:, 7] = 1.0, 1e20
    res = r.render_ray_queries(q)
    assert (res[:, 0] == -1).all() and (res[:, 2].view(np.int32) == -1).all()
    r.close()


# ---------------------------------------------------------------- tail kernel (late bounces in one launch)
@pytest.mark.parametrize("scene_name,variant", [("two_level_test", abi.VARIANT_GLTF), ("alpha_test", abi.VARIANT_GLTF), ("cornell32", abi.VARIANT_SIMPLE)])
def test_tail_kernel_is_bit_identical_to_the_standalone_launches(scene_name, variant):
    """rp_k_tail runs extend / shade / connect of the late bounces on block-local lists: same device code, same bits, wherever
    the hand-over happens (fixed bounce 1..4, or chosen from the previous frame's queue lengths)"""
    import os
    s = getattr(scenes, scene_name)()
    W, H, spp = 96, 72, 3
    images = {}
    for mode in ("0", "1", "2", "4", "-1"):
        os.environ["RPTR_TAIL_BOUNCE"] = mode
        os.environ["RPTR_TAIL_THRESHOLD"] = "100000000"     # adaptive: hand over as early as it may
        try:
            r = backend.RenderHip()
            r.initialize(W, H)
            r.set_scene(s)
            frames = []
            for k in range(3):                               # three accumulating frames: the adaptive mode settles after the first
                img, st, _ = gpu_render(s, W, H, spp, variant, reset=(k == 0), renderer=r)
                frames.append((img.copy(), int(st.raw.launches_extend), int(st.raw.rays_closest), int(st.raw.rays_shadow)))
            r.close()
        finally:
            del os.environ["RPTR_TAIL_BOUNCE"], os.environ["RPTR_TAIL_THRESHOLD"]
        images[mode] = frames
    depth = abi.RenderParams.default().max_path_depth
    assert [f[1] for f in images["0"]] == [depth] * 3 and [f[1] for f in images["2"]] == [2] * 3
    assert images["-1"][0][1] == depth and images["-1"][2][1] == 1          # first frame without a tail, then from bounce 1
    for mode in ("1", "2", "4", "-1"):
        for a, b in zip(images["0"], images[mode]):
            assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), mode
            assert a[2:] == b[2:], mode                                      # the same rays were traced


# ---------------------------------------------------------------- flattened instances (RPTR_FLATTEN)
@pytest.mark.library_defaults
@pytest.mark.parametrize("scene_name", ["two_level_test", "alpha_test"])
def test_flattened_scene_matches_the_oracle_on_the_same_tree(scene_name):
    """THE LIBRARY'S DEFAULTS (no option, no environment variable: ADVICE r5 -- the configuration a host that sets nothing gets) on
    multi-instance scenes with several parameterized meshes per mesh (two_level_test: instances that resolve their materials through their
    own geometry records) and with alpha-tested materials (alpha_test): one world-space tree over all instanced triangles; every ray walks
    it like the oracle does (results and visit counts), the image agrees with the oracle on that tree, hits name the right instance, primitive
    and material (shading reads the mesh streams through the instance record), and the two-level walk of the same scene is matched up to the
    rounding of the pre-transformed triangles"""
    s = getattr(scenes, scene_name)()
    W, H, spp = 128, 96, 2
    img, st, r = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, keep=True, count=True)
    assert r.get_option("flatten") == -1 and r.get_option("fast_math") == 0        # nothing was set
    assert bool(np.frombuffer(np.ascontiguousarray(r.export_bvh()[2]).tobytes(), np.int32).reshape(-1, 32)[0, 15] & 1)   # RPTR_BVH_INSTANCE_FLAT
    osc = _oracle_on_device_tree(s, r)
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, bvh_mode=O.BVH_IMPORTED, count=True)
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL
    assert abs(int(st.raw.rays_closest) - ost.rays_closest) <= 1e-3 * ost.rays_closest
    assert abs(int(st.raw.nodes_closest) - ost.nodes_closest) <= 1e-3 * ost.nodes_closest
    if scene_name == "two_level_test":
        assert_ray_visit_parity(r, osc, 64, 48, 1, abi.VARIANT_GLTF)
    # ray queries against the two-level brute force: same triangles, t / u / v up to rounding
    q = random_queries(np.random.default_rng(5), 20000, -3, 3)
    res = r.render_ray_queries(q)
    brute = np.zeros_like(res)
    O.OracleScene(s).trace(q, bvh_mode=O.BVH_BRUTE, out=brute)
    hit = brute[:, 0] >= 0
    agree = (res[:, 2:].view(np.int32) == brute[:, 2:].view(np.int32)).all(axis=1)
    assert hit.sum() > 500 and agree.mean() > 0.9995
    both = hit & agree
    assert np.allclose(res[both, :2], brute[both, :2], atol=2e-4)
    r.close()
    # and the scene rendered through the two-level tree looks the same
    img2, _, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, options={"flatten": 0})
    if scene_name == "alpha_test":     # fractional alphas draw from the path's generator in candidate order: another tree, other draws
        assert abs(float(np.nanmean(img[..., :3])) - float(np.nanmean(img2[..., :3]))) < 0.03 * float(np.nanmean(img2[..., :3]))
    else:
        assert image_error(img, img2)[0] < 5 * RMSE_TOL


def test_deep_traversal_stacks_spill_to_global_memory_and_stay_exact():
    """The traversal keeps 20 stack entries per lane in LDS and spills deeper ones to a per-thread slice of global memory (csrc/dtraverse.h: the
    whole wave takes the generic push / pop path once a lane comes within three entries of the end). No other scene of the suite gets there often
    (tools/soak_stack_spill.sh forces it with a four-entry build); this one does on the shipped library: 65536 parallel pages, a tree of depth >= 8
    whose every node has four children a crossing ray hits -- 3 x depth + 1 entries before the first triangle test. Results bit-equal to the
    oracle's walk of the same tree, node / triangle visits of every closest-hit and occlusion ray of a frame equal to the oracle's walk of the exported tree."""
    s = scenes.book(65536)
    r = backend.RenderHip()
    r.initialize(64, 48)
    r.set_scene(s)
    nodes, _, insts = r.export_bvh()[:3]
    child = np.frombuffer(np.ascontiguousarray(nodes).tobytes(), np.int32).reshape(-1, 16)[:, 10:14]
    root = int(np.frombuffer(np.ascontiguousarray(insts).tobytes(), np.int32).reshape(-1, 32)[0, 12])
    depth, level = 0, [root]
    while level:
        depth += 1
        level = [int(c) for n in level for c in child[n] if c >= 0]
    assert 3 * (depth - 1) + 1 > 20, depth          # the first descent alone parks more than the LDS part holds
    rng = np.random.default_rng(11)
    q = np.zeros((20000, 8), np.float32)
    q[:, 0:2] = rng.uniform(-0.6, 0.6, (len(q), 2))
    q[:, 2] = np.where(rng.random(len(q)) < 0.5, -3.0, 3.0)
    d = np.concatenate([rng.normal(scale=0.05, size=(len(q), 2)), -np.sign(q[:, 2:3])], axis=1)
    q[:, 4:7] = d / np.linalg.norm(d, axis=1, keepdims=True)
    q[:, 7] = 1e20
    res = np.full((len(q), 4), 3.0, np.float32)
    r.render_ray_queries(q, res)
    osc = O.OracleScene(s)
    osc.import_bvh(*r.export_bvh())
    ref = np.full((len(q), 4), 3.0, np.float32)
    osc.trace(q, bvh_mode=O.BVH_IMPORTED, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32))           # the same tree: every bit
    assert (res[:, 0] >= 0).sum() > 10000
    # another tree (the oracle's own, float boxes) may disagree on a ray that leaves the book exactly through a page's outer edge, where the
    # triangle's edge IS its box's face (one or two rays in 20000 name the neighbouring page): everywhere else the hit does not depend on the tree
    own = np.full((len(q), 4), 3.0, np.float32)
    osc.trace(q, bvh_mode=O.BVH_OWN, out=own)
    assert (res.view(np.uint32) == own.view(np.uint32)).all(axis=1).mean() > 0.9995
    assert_ray_visit_parity(r, osc, 64, 48, 1, abi.VARIANT_SIMPLE)
    r.close()


def test_reinitialize_and_new_scene_on_one_handle():
    """initialize() and set_scene() may come again (resize, scene switch: app.cpp:445,150-175): old buffers are released, the
    adaptive tail hand-over starts over, images stay right and the reported device memory is what is allocated now"""
    r = backend.RenderHip(frames_in_flight=2)
    s1, s2 = scenes.textured_test(), scenes.two_level_test()
    reported = {}
    for W, H, s in [(64, 48, s1), (128, 96, s2), (64, 48, s1)]:
        r.initialize(W, H)
        r.set_scene(s)
        for k in range(3):
            st = r.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=(k == 0)), spp=2)
        img = np.zeros((H, W, 4), np.float32)
        assert r.readback_framebuffer(img) == W * H * 4
        ref, _ = O.OracleScene(s).render(W, H, 6)
        rmse, same, _ = image_error(img, ref)
        assert same and rmse < RMSE_TOL
        reported.setdefault((W, H, s.name), []).append(int(st.raw.device_bytes_allocated))
    first, again = reported[(64, 48, s1.name)]
    assert first == again
    r.close()


def test_full_size_forest_flattened_against_two_level():
    """BASELINE configs[3] at full size (10 M instanced triangles, 1080p): the world-space tree and the two-level tree trace the same
    frame -- same closest-hit queries up to silhouette flips, images equal up to those pixels -- the flattened one with fewer node
    fetches; its stripe split is bit-identical to the whole frame; a band of rows agrees with the oracle's own two-level tree"""
    import os
    s = scenes.forest()
    W, H = 1920, 1080
    two, st2, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF, count=True)
    os.environ["RPTR_FLATTEN"] = "1"
    try:
        flat, st1, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF, count=True)
        half = np.zeros_like(flat)
        for rank in range(2):
            img, _, _ = gpu_render(s, W, H, 1, abi.VARIANT_GLTF, rank=rank, world=2, stripe_rows=8)
            rows = [y for y in range(H) if (y // 8) % 2 == rank]
            half[rows] = img[rows]
    finally:
        del os.environ["RPTR_FLATTEN"]
    assert np.array_equal(flat.view(np.uint32), half.view(np.uint32))
    assert abs(int(st1.raw.rays_closest) - int(st2.raw.rays_closest)) <= 2e-3 * int(st2.raw.rays_closest)
    assert st1.raw.nodes_closest < 0.8 * st2.raw.nodes_closest
    differ = (np.abs(flat[..., :3] - two[..., :3]).max(axis=2) > 1e-4).mean()
    assert differ < 0.02                                           # a path that flips at a silhouette changes its whole pixel
    # the oracle's OWN two-level tree differs from the flattened walk at silhouettes (world-space vs object-space triangles): a smoke bound
    rows = (600, 604)
    osc = O.OracleScene(s)
    ref, _ = osc.render(W, H, 1, variant=abi.VARIANT_GLTF, rows=rows)
    band = np.abs(flat[rows[0]:rows[1], :, :3] - ref[rows[0]:rows[1], :, :3]).max(axis=2)
    assert (band > 1e-4).mean() < 0.02
    # the contract (RMSE < 1e-3) holds against the oracle walking the very tree the benchmark times: the exported flattened one
    os.environ["RPTR_FLATTEN"] = "1"
    try:
        r = backend.RenderHip()
        r.initialize(64, 64)
        r.set_scene(s)
        osc.import_bvh(*r.export_bvh())
        r.close()
    finally:
        del os.environ["RPTR_FLATTEN"]
    for rows in ((420, 424), (600, 604), (880, 884)):
        ref, _ = osc.render(W, H, 1, variant=abi.VARIANT_GLTF, rows=rows, bvh_mode=O.BVH_IMPORTED)
        rmse, same, maxabs = image_error(flat[rows[0]:rows[1]], ref[rows[0]:rows[1]])
        assert same and rmse < RMSE_TOL, (rows, rmse, maxabs)


def test_raster_taa_screen_jitter_matches_the_oracle():
    """render_params.enable_raster_taa: no pixel-filter draw, all primary rays of a frame share view_params.screen_jitter = entry
    (frame_offset + frame_id) % 16 of the (2, 3) Halton table (pt_megakernel.glsl:316-320, render_vulkan.cpp:2917-2926), which the
    motion / jitter AOV carries in zw (accumulate.glsl:85). Three frames: reset, accumulate, reset -> entries 0, 2 and 4."""
    s = scenes.textured_test()
    W, H, spp = 96, 80, 2
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    r.params.enable_raster_taa = 1
    p = abi.RenderParams.default()
    p.enable_raster_taa = 1
    osc = O.OracleScene(s)
    cam = s.camera_params()
    plan = [(True, 0, 0), (False, 2, 0), (True, 0, 4)]   # (reset, samples accumulated before = frame_id, frame_offset)
    ref = None
    images = []
    for reset, frame_id, frame_offset in plan:
        st = r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=reset), spp=spp)
        img = np.zeros((H, W, 4), np.float32)
        r.readback_framebuffer(img)
        ref, _, aovs = osc.render(W, H, spp, params=p, sample_begin=frame_id, frame_offset=frame_offset, aovs=True,
                                  accum=None if reset else ref.copy())
        rmse, same, _ = image_error(img, ref)
        assert same and rmse < RMSE_TOL, (frame_id, frame_offset, rmse)
        got = np.zeros((H, W, 4), np.float16)
        r.readback_aov(r.AOVMotionJitterIndex, got)
        h = O.halton23((frame_offset + frame_id) % 16)
        want = np.array([h[0] * 2 / W - 1 / W, h[1] * 2 / H - 1 / H], np.float32).astype(np.float16)
        assert np.array_equal(got[..., 2:].reshape(-1, 2), np.broadcast_to(want, (W * H, 2))) and np.array_equal(got[..., 2:], aovs[2][..., 2:])
        images.append(img)
    # the jitter moved the image: without TAA-off pixel filtering two 2-spp frames of a static view differ only through it and the seeds
    assert not np.array_equal(images[0], images[2])
    r.close()


def test_reprojection_mode_discard_history():
    """render_params.reprojection_mode = REPROJECTION_MODE_DISCARD_HISTORY (process_samples.comp:116-131): the accumulation buffer of a
    frame that does not reset holds that frame's samples only (seeded as samples 2..3), not the running mean over the history"""
    s = scenes.cornell32()
    W, H, spp = 64, 64, 2
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(s)
    r.params.reprojection_mode = 1
    p = abi.RenderParams.default()
    p.reprojection_mode = 1
    osc = O.OracleScene(s)
    cam = s.camera_params()
    first = np.zeros((H, W, 4), np.float32)
    second = np.zeros((H, W, 4), np.float32)
    r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=spp)
    r.readback_framebuffer(first)
    r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=False), spp=spp)
    r.readback_framebuffer(second)
    ref1, _ = osc.render(W, H, spp, params=p)
    ref2, _ = osc.render(W, H, spp, params=p, sample_begin=spp, accum=ref1.copy())
    keep, _ = osc.render(W, H, spp, sample_begin=spp, accum=ref1.copy())      # the default mode folds the history in
    for got, ref in ((first, ref1), (second, ref2)):
        rmse, same, _ = image_error(got, ref)
        assert same and rmse < RMSE_TOL
    assert image_error(second, keep)[0] > 10 * RMSE_TOL
    r.close()


def test_shadow_rays_on_the_side_stream_change_no_bit(monkeypatch):
    """a handle with ONE frame context runs connect(b) on a side stream beside extend(b + 1) (csrc/rptr_hip.hip: side_connect, the default for
    frames_in_flight = 1): the order of float additions into a path's radiance is kept by the joins, so the image, the ray counts and the
    counted traversal work equal those of the single-stream schedule bit for bit"""
    s = scenes.grid(120, 60, with_emitters=True)
    W, H, spp = 160, 96, 3
    out = []
    for side in ("0", "1", None):
        if side is None:
            monkeypatch.delenv("RPTR_SIDE_CONNECT", raising=False)
        else:
            monkeypatch.setenv("RPTR_SIDE_CONNECT", side)
        img, st, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, count=True)
        out.append((img, int(st.raw.rays_closest), int(st.raw.rays_shadow), int(st.raw.nodes_visited), int(st.raw.tris_tested)))
    for other in out[1:]:
        assert np.array_equal(out[0][0].view(np.uint32), other[0].view(np.uint32)) and out[0][1:] == other[1:]
    assert out[0][2] > 0


def test_discard_history_keeps_all_samples_of_a_frame_the_backend_splits(monkeypatch):
    """a frame of more samples than fit in flight is rendered in several internal launches (RPTR_MAX_BATCH_SPP): with
    REPROJECTION_MODE_DISCARD_HISTORY the displayed frame still holds ALL of its samples (only earlier frames are dropped), bit for bit the
    image of the same frame rendered in one launch"""
    s = scenes.cornell32()
    W, H, spp = 48, 48, 4
    cam = s.camera_params()
    images = []
    for split in (None, "1", "3"):
        if split:
            monkeypatch.setenv("RPTR_MAX_BATCH_SPP", split)
        r = backend.RenderHip()
        r.initialize(W, H)
        r.set_scene(s)
        r.params.reprojection_mode = 1
        img = np.zeros((H, W, 4), np.float32)
        r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=2)   # history to discard
        r.render(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=False), spp=spp)
        r.readback_framebuffer(img)
        images.append(img)
        r.close()
        if split:
            monkeypatch.delenv("RPTR_MAX_BATCH_SPP")
    assert np.array_equal(images[0], images[1]) and np.array_equal(images[0], images[2])
    p = abi.RenderParams.default()
    p.reprojection_mode = 1
    osc = O.OracleScene(s)
    ref1, _ = osc.render(W, H, 2, params=p)
    ref2, _ = osc.render(W, H, spp, params=p, sample_begin=2, accum=ref1.copy())
    rmse, same, _ = image_error(images[0], ref2)
    assert same and rmse < RMSE_TOL


def _with_mip_chains(s):
    """box-filtered mip chains for every texture of the scene (test data: the product generates none)"""
    for t in s.textures:
        lv, cur = [], np.asarray(t.rgba).astype(np.float64)
        while cur.shape[0] > 1 or cur.shape[1] > 1:
            h, w = max(1, cur.shape[0] // 2), max(1, cur.shape[1] // 2)
            cur = cur[:2 * h if cur.shape[0] > 1 else 1, :2 * w if cur.shape[1] > 1 else 1]
            cur = cur.reshape(h, cur.shape[0] // h, w, cur.shape[1] // w, 4).mean(axis=(1, 3))
            lv.append(np.clip(np.round(cur), 0, 255).astype(np.uint8))
        t.mips = lv or None
    return s


@pytest.mark.parametrize("variant", [abi.VARIANT_GLTF, abi.VARIANT_SIMPLE])
def test_mip_mapped_textures_and_path_footprints(variant):
    """textures with mip levels, seen small and at a grazing angle: the footprint a path carries (pt_megakernel.glsl:336-352, 582-606,
    698-702) picks level and anisotropy of every material lookup (textureGrad) and the normal map drops a level per bounce; image within
    tolerance of the oracle at two resolutions, and the levels do matter (the same scene without them renders another image)"""
    s = _with_mip_chains(scenes.textured_test())
    cam = s.camera_params()
    for k in range(3):                                   # step back and down: minification, anisotropy
        cam.pos[k] = cam.pos[k] - 2.5 * cam.dir[k]
    cam.pos[1] -= 0.6
    osc = O.OracleScene(s)
    for W, H, spp in ((96, 72, 3), (240, 180, 1)):
        r = backend.RenderHip()
        r.initialize(W, H)
        r.set_scene(s)
        r.render(backend.RenderConfiguration(cam, active_variant=variant, reset_accumulation=True), spp=spp)
        img = np.zeros((H, W, 4), np.float32)
        r.readback_framebuffer(img)
        r.close()
        ref, _ = osc.render(W, H, spp, variant=variant, camera=cam)
        rmse, same, maxabs = image_error(img, ref)
        assert same and rmse < RMSE_TOL, (W, H, rmse, maxabs)
    flat = scenes.textured_test()
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(flat)
    r.render(backend.RenderConfiguration(cam, active_variant=variant, reset_accumulation=True), spp=spp)
    img2 = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img2)
    r.close()
    assert image_error(img, img2)[0] > 10 * RMSE_TOL
