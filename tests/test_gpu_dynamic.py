"""Dynamic meshes: rptr_hip_update_vertices + rptr_hip_refit against the oracle.

The reference updates the BLAS of a dynamic mesh in place and refits the TLAS
(render_vulkan.cpp:942-952,1323-1354) and shades dynamic geometry from the float
vertex buffer (pt_megakernel.glsl:526-529). The device refit keeps the topology
and recomputes triangles and boxes; the oracle REBUILDS its own tree from the new
float positions. Closest-hit results are defined independently of the topology
(min t, ties by ids), so refit-then-trace must equal rebuild-then-trace bit for bit.
"""
import numpy as np
import pytest

import oracle_lib as O
from common import RMSE_TOL, assert_ray_visit_parity, gpu_render, image_error, random_queries
from realtimepathtracingresearchframework_amd import abi, backend, scenes

pytestmark = pytest.mark.gpu

NX, NZ = 96, 48


@pytest.fixture(scope="module")
def dyn_grid():
    return scenes.grid(NX, NZ, deform_t=0.0, name="dyn-grid")


def _grid_queries(n, seed):
    q = random_queries(np.random.default_rng(seed), n, -60, 60)
    q[:, 1] = np.abs(q[:, 1]) * 0.2 + 3.0
    q[:, 5] = -np.abs(q[:, 5])
    return q


def test_refit_of_unchanged_vertices_reproduces_the_built_tree(dyn_grid):
    """update with the dequantised positions + refit == the tree set_scene built, bit for bit (boxes and triangles)."""
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(dyn_grid)
    n0, t0, i0 = (a.copy() for a in r.export_bvh())
    g = dyn_grid.geometries[0]
    r.update_vertices(0, scenes.dequantize_positions(g.qpos, g.scaling, g.offset))
    r.refit()
    n1, t1, i1 = r.export_bvh()
    assert np.array_equal(n0.view(np.uint32), n1.view(np.uint32))
    assert np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    assert np.array_equal(i0.view(np.uint32), i1.view(np.uint32))
    r.close()


@pytest.mark.parametrize("t", [0.3, 0.65])
def test_refit_then_trace_equals_rebuild_then_trace(dyn_grid, t):
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(dyn_grid)
    P = scenes.grid_positions(NX, NZ, t)
    q = _grid_queries(20000, 11)
    before = r.render_ray_queries(q).copy()
    r.update_vertices(0, P)
    r.refit()
    res = r.render_ray_queries(q)
    osc = O.OracleScene(dyn_grid)
    osc.set_dynamic_vertices(0, P)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_OWN, out=ref)        # oracle: fresh SAH build over the new positions
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32))
    assert (res[:, 0] >= 0).mean() > 0.2 and not np.array_equal(res, before)
    # every box of the refitted tree still bounds its subtree: walking the exported tree finds the same hits and
    # the oracle's visit counters equal the device's
    osc.import_bvh(*r.export_bvh())
    ref2 = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_IMPORTED, out=ref2)
    assert np.array_equal(res.view(np.uint32), ref2.view(np.uint32))
    r.close()


def test_image_parity_after_refit(dyn_grid):
    """shading reads the float vertex buffer of a dynamic geometry (normals/uvs stay the quantised stream)."""
    W, H, spp = 160, 90, 2
    r = backend.RenderHip()
    r.initialize(W, H)
    r.set_scene(dyn_grid)
    P = scenes.grid_positions(NX, NZ, 0.4)
    r.update_vertices(0, P)
    r.refit()
    img, st, _ = gpu_render(dyn_grid, W, H, spp, abi.VARIANT_GLTF, renderer=r, count=True)
    osc = O.OracleScene(dyn_grid)
    osc.set_dynamic_vertices(0, P)
    osc.import_bvh(*r.export_bvh())
    ref, ost = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, bvh_mode=O.BVH_IMPORTED, count=True)
    rmse, same, _ = image_error(img, ref)
    assert same and rmse < RMSE_TOL
    assert st.raw.rays_closest == ost.rays_closest and st.raw.rays_shadow == ost.rays_shadow
    assert abs(int(st.raw.nodes_visited) - (ost.nodes_closest + ost.nodes_shadow)) <= 1e-4 * st.raw.nodes_visited
    assert abs(int(st.raw.tris_tested) - (ost.tris_closest + ost.tris_shadow)) <= 1e-4 * st.raw.tris_tested
    assert_ray_visit_parity(r, osc, W, H, spp, abi.VARIANT_GLTF)
    # and against the oracle's own rebuilt tree (different topology, same image)
    ref2, _ = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, bvh_mode=O.BVH_OWN)
    assert image_error(img, ref2)[0] < RMSE_TOL
    r.close()


def test_refit_of_an_instanced_dynamic_mesh_updates_the_top_level():
    """12 instances (rotation + scale) of two meshes; mesh 1 is dynamic and grows: instance boxes + TLAS must follow."""
    s = scenes.two_level_test()
    s.meshes[1].dynamic = True
    r = backend.RenderHip()
    r.initialize(96, 64)
    r.set_scene(s)
    g = s.geometries[1]
    P0 = scenes.dequantize_positions(g.qpos, g.scaling, g.offset)
    P = (P0 * np.float32(1.7) + np.array([0.5, -0.25, 0.3], np.float32)).astype(np.float32)
    r.update_vertices(1, P)
    r.refit()
    q = random_queries(np.random.default_rng(4), 20000, -6, 6)
    res = r.render_ray_queries(q)
    osc = O.OracleScene(s)
    osc.set_dynamic_vertices(1, P)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32))
    assert (res[:, 0] >= 0).sum() > 500
    # image: NaN masks agree too (this scene holds the reference's degenerate coplanar-light sample)
    img, _, _ = gpu_render(s, 96, 64, 2, abi.VARIANT_GLTF, renderer=r)
    ref_img, _ = osc.render(96, 64, 2, variant=abi.VARIANT_GLTF)
    rmse, same, _ = image_error(img, ref_img)
    assert same and rmse < RMSE_TOL
    r.close()


def test_repeated_updates_do_not_drift(dyn_grid):
    """animate t = 0.1 .. 0.5 with a refit per frame; the final state equals a single update to t = 0.5."""
    q = _grid_queries(5000, 3)
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(dyn_grid)
    for t in (0.1, 0.2, 0.3, 0.4, 0.5):
        r.update_vertices(0, scenes.grid_positions(NX, NZ, t))
        r.refit()
    a = r.render_ray_queries(q).copy()
    na = [x.copy() for x in r.export_bvh()]
    r.close()
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(dyn_grid)
    r.update_vertices(0, scenes.grid_positions(NX, NZ, 0.5))
    r.refit()
    b = r.render_ray_queries(q)
    nb = r.export_bvh()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert all(np.array_equal(x.view(np.uint32), y.view(np.uint32)) for x, y in zip(na, nb))
    r.close()


def test_update_vertices_error_convention(dyn_grid):
    static = scenes.grid(16, 8)
    r = backend.RenderHip()
    r.initialize(32, 32)
    with pytest.raises(backend.BackendError):   # before set_scene
        r.update_vertices(0, np.zeros((3, 3), np.float32))
    r.set_scene(static)
    with pytest.raises(backend.BackendError) as e:   # geometry of a static mesh
        r.update_vertices(0, np.zeros((16 * 8 * 6, 3), np.float32))
    assert e.value.code == abi.RPTR_E_INVALID
    r.refit()                                        # nothing dirty: a no-op, not an error
    r.set_scene(dyn_grid)
    with pytest.raises(backend.BackendError):        # wrong vertex count
        r.update_vertices(0, np.zeros((5, 3), np.float32))
    with pytest.raises(backend.BackendError):        # geometry out of range
        r.update_vertices(7, np.zeros((3, 3), np.float32))
    r.close()


def test_device_source_update_equals_host_source_update(dyn_grid):
    """rptr_hip_update_vertices_device: the animation lives on the GPU (torch tensor here, a compute shader in the reference)."""
    import torch
    P = scenes.grid_positions(NX, NZ, 0.25)
    q = _grid_queries(8000, 9)
    out = []
    for device_src in (False, True):
        r = backend.RenderHip(stream=torch.cuda.current_stream().cuda_stream)
        r.initialize(64, 64)
        r.set_scene(dyn_grid)
        if device_src:
            t = torch.from_numpy(P).cuda()
            r.update_vertices_device(0, t.data_ptr(), t.shape[0])
        else:
            r.update_vertices(0, P)
        r.refit()
        out.append((r.render_ray_queries(q).copy(), [x.copy() for x in r.export_bvh()]))
        r.close()
    assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
    assert all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(out[0][1], out[1][1]))


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_refit_of_soups(seed):
    """every mesh dynamic; vertices jump far, some triangles collapse to points, one mesh becomes flat: refit (topology of the
    ORIGINAL build) must still give brute-force answers, with sheared / mirrored instances on top"""
    s = scenes.soup(seed)
    for m in s.meshes:
        m.dynamic = True
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    osc = O.OracleScene(s)
    rng = np.random.default_rng(seed)
    for gi, g in enumerate(s.geometries):
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset)
        P = (P + rng.normal(size=P.shape) * 0.3).astype(np.float32)      # far from where the tree was built
        P[0:3] = P[0]                                                      # a triangle collapsed to a point
        if gi == 0:
            P[:, 2] = 0.5                                                  # the whole geometry flattened
        r.update_vertices(gi, P)
        osc.set_dynamic_vertices(gi, P)
    r.refit()
    q = random_queries(np.random.default_rng(seed + 1), 20000, -6, 6)
    res = r.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).sum() > 500
    assert_ray_visit_parity(r, osc, 64, 64, 1, abi.VARIANT_GLTF)
    r.close()


def test_animated_frames_in_flight_equal_one_at_a_time(dyn_grid):
    """3 frame contexts on a dynamic scene: every context keeps its own tree and float vertices, brought up to date when a
    frame is submitted on it. A sequence (update, refit, frame)x5 queued back to back gives the same images as the same
    sequence rendered one frame at a time -- also when only every other frame is preceded by an update."""
    W, H = 128, 72
    times = [0.1, None, 0.35, 0.35, 0.6]        # None: no update before this frame (static continuation)

    def run(fif):
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(dyn_grid)
        cam = dyn_grid.camera_params()
        images, queue = [], []
        def collect():
            r.wait(queue.pop(0))
            img = np.zeros((H, W, 4), np.float32)
            r.readback_framebuffer(img)
            images.append(img)
        for t in times:
            if t is not None:
                r.update_vertices(0, scenes.grid_positions(NX, NZ, t))
                r.refit()
            cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
            queue.append(r.render_async(cfg, spp=1))
            if len(queue) >= fif:
                collect()
        while queue:
            collect()
        # ray queries work on the handle's own (master) tree: current after the last refit
        q = _grid_queries(4000, 5)
        res = r.render_ray_queries(q).copy()
        r.close()
        return images, res

    ref_images, ref_q = run(1)
    images, q3 = run(3)
    for a, b in zip(images, ref_images):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.array_equal(q3.view(np.uint32), ref_q.view(np.uint32))
    assert not np.array_equal(ref_images[0], ref_images[2])     # the surface really moved
    # reset_accumulation re-seeds every frame (frame_offset advances), so even frames of the same geometry differ in noise
    assert not np.array_equal(ref_images[2], ref_images[3])
    # oracle check of the last frame: geometry of t = 0.6, frame_offset = 4 resets so far
    osc = O.OracleScene(dyn_grid)
    osc.set_dynamic_vertices(0, scenes.grid_positions(NX, NZ, 0.6))
    ref, _ = osc.render(W, H, 1, variant=abi.VARIANT_SIMPLE, frame_offset=4)
    rmse, same, _ = image_error(images[-1], ref)
    assert same and rmse < RMSE_TOL


# ---------------------------------------------------------------- device-side rebuild + BVH policy (csrc/lbvh.h)
def _rebuild_check(r, osc, s, seed, lo=-60, hi=60):
    """ray queries through the device-built tree == the oracle's brute force / own tree, bit for bit; every ray of a frame walks the
    exported tree exactly as the oracle does (same nodes, same triangles)"""
    q = _grid_queries(20000, seed) if lo == -60 else random_queries(np.random.default_rng(seed), 20000, lo, hi)
    res = r.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_OWN, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).sum() > 500
    assert_ray_visit_parity(r, osc, 64, 48, 1, abi.VARIANT_SIMPLE)


@pytest.mark.parametrize("nx,nz", [(96, 48), (400, 200)])
def test_forced_device_rebuild_equals_refit_and_oracle(nx, nz):
    """force_bvh_rebuild: every refit of changed vertices builds a new tree on the device (Morton order, binary radix tree, 4-wide
    collapse, shared encoder). Hits equal those of the refitted host tree and of the oracle's rebuild; the exported tree is walked
    identically by the oracle; a later plain refit of the device-built tree (bottom-up, node count read on the device) is right too."""
    s = scenes.grid(nx, nz, deform_t=0.0, name="dyn-grid-rebuild")
    r = backend.RenderHip()
    r.initialize(64, 48)
    r.set_scene(s)
    osc = O.OracleScene(s)
    # 1. refit only (no policy): the reference result for the same vertices
    P1 = scenes.grid_positions(nx, nz, 0.3)
    r.update_vertices(0, P1)
    r.refit()
    q = _grid_queries(20000, 3)
    refit_hits = r.render_ray_queries(q).copy()
    assert r.bvh_rebuild_count() == 0
    # 2. the same vertices again, now with a forced rebuild
    r.set_bvh_policy(force_bvh_rebuild=True)
    r.update_vertices(0, P1)
    r.refit()
    assert r.bvh_rebuild_count() == 1
    assert np.array_equal(r.render_ray_queries(q).view(np.uint32), refit_hits.view(np.uint32))
    osc.set_dynamic_vertices(0, P1)
    _rebuild_check(r, osc, s, 5)
    # 3. policy off again: new vertices, a bottom-up refit of the DEVICE-built tree
    r.set_bvh_policy()
    P2 = scenes.grid_positions(nx, nz, 0.8)
    r.update_vertices(0, P2)
    r.refit()
    assert r.bvh_rebuild_count() == 1
    osc.set_dynamic_vertices(0, P2)
    _rebuild_check(r, osc, s, 6)
    # 4. refitting unchanged vertices reproduces the device-built tree bit for bit
    n0, t0, _ = (a.copy() for a in r.export_bvh())
    r.update_vertices(0, P2)
    r.refit()
    n1, t1, _ = r.export_bvh()
    assert np.array_equal(n0.view(np.uint32), n1.view(np.uint32)) and np.array_equal(t0.view(np.uint32), t1.view(np.uint32))
    # image of the rebuilt scene against the oracle
    img, _, _ = gpu_render(s, 64, 48, 2, abi.VARIANT_SIMPLE, renderer=r)
    ref, _ = osc.render(64, 48, 2, variant=abi.VARIANT_SIMPLE, frame_offset=0)
    assert image_error(img, ref)[0] < RMSE_TOL
    r.close()


def test_rebuild_triangle_budget_spreads_rebuilds_over_refits():
    """rebuild_triangle_budget = half the mesh: a new tree every second refit call, refits in between; images of an animated
    sequence equal those of the refit-only run within tolerance (same hits, different tree), also with frames in flight"""
    s = scenes.grid(NX, NZ, deform_t=0.0, name="dyn-grid-budget")
    ntri = s.num_tris()
    W, H = 96, 64
    times = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6]

    def run(budget, fif):
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(s)
        r.set_bvh_policy(rebuild_triangle_budget=budget)
        images, queue = [], []

        def collect():
            r.wait(queue.pop(0))
            img = np.zeros((H, W, 4), np.float32)
            r.readback_framebuffer(img)
            images.append(img)
        for t in times:
            r.update_vertices(0, scenes.grid_positions(NX, NZ, t))
            r.refit()
            queue.append(r.render_async(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True), spp=1))
            if len(queue) >= fif:
                collect()
        while queue:
            collect()
        n = r.bvh_rebuild_count()
        r.close()
        return images, n

    ref_images, n0 = run(0, 1)
    images, n1 = run(ntri // 2, 1)
    assert n0 == 0 and n1 == len(times) // 2
    for a, b in zip(images, ref_images):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))   # the closest hit does not depend on the tree: same paths, same image
    images3, n3 = run(ntri // 2, 3)
    assert n3 >= len(times) // 2                                       # (every frame context rebuilds its own copy)
    for a, b in zip(images3, ref_images):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("seed", [31, 32])
def test_device_rebuild_of_soups_and_tiny_meshes(seed):
    """degenerate input for the builder: meshes of a few hundred triangles with collapsed / flat geometry, instanced with shears
    and mirrors, every mesh dynamic and rebuilt on the device"""
    s = scenes.soup(seed, tris_per_mesh=37)
    for m in s.meshes:
        m.dynamic = True
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    r.set_bvh_policy(force_bvh_rebuild=True)
    osc = O.OracleScene(s)
    rng = np.random.default_rng(seed)
    for gi, g in enumerate(s.geometries):
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset)
        P = (P + rng.normal(size=P.shape) * 0.2).astype(np.float32)
        P[0:3] = P[0]
        if gi == 1:
            P[:] = P[0]                                                     # a whole geometry collapsed to a point: all centroids equal
        r.update_vertices(gi, P)
        osc.set_dynamic_vertices(gi, P)
    r.refit()
    assert r.bvh_rebuild_count() == len(s.meshes)
    q = random_queries(np.random.default_rng(seed + 1), 20000, -6, 6)
    res = r.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).sum() > 200
    assert_ray_visit_parity(r, osc, 64, 64, 1, abi.VARIANT_GLTF)
    r.close()


def test_device_built_tree_is_as_good_as_the_host_tree_on_a_height_field():
    """quality of the device-side rebuild: on a flat mesh the Morton cells must be cubic (one scale for the three axes) -- with per-axis
    scaling the curve split the height field by height and a ray visited 50 % more nodes. Node visits per closest-hit ray of a frame on
    the device-built tree stay within 10 % of the host's binned-SAH tree."""
    nx, nz = 300, 150
    s = scenes.grid(nx, nz, deform_t=0.0, name="dyn-grid-quality")
    r = backend.RenderHip()
    r.initialize(320, 180)
    r.set_scene(s)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
    P = scenes.grid_positions(nx, nz, 0.0)
    visits = {}
    for label, force in (("host", False), ("device", True)):
        r.set_bvh_policy(force_bvh_rebuild=force)
        r.update_vertices(0, P)
        r.refit()
        st = r.render(cfg, spp=2, count_traversal=True).raw
        visits[label] = st.nodes_closest / st.rays_closest
    assert r.bvh_rebuild_count() == 1
    assert visits["device"] < 1.10 * visits["host"], visits
    r.close()


def test_rebuild_of_dynamic_meshes_with_many_instances():
    """scenes with 16 or more instances open the roots of their bottom-level trees in the top level (instance records name sub-roots):
    not for dynamic meshes -- a device-side rebuild changes the topology under those records (geometry used to vanish). 40 rotated,
    scaled instances of two dynamic tree meshes: refit and rebuild give the hits of the static scene, bit for bit"""
    s = scenes.forest(n_meshes=2, tris_per_tree=600, n_instances=40, name="dyn-forest")
    static_hits = None
    q = random_queries(np.random.default_rng(9), 60000, -12, 12)
    q[:, 1] = np.abs(q[:, 1]) * 0.2 + 0.1
    r0 = backend.RenderHip()
    r0.initialize(32, 32)
    r0.set_scene(s)
    static_hits = r0.render_ray_queries(q).copy()
    r0.close()
    assert (static_hits[:, 0] >= 0).sum() > 2000
    for m in s.meshes[:-1]:
        m.dynamic = True
    r = backend.RenderHip()
    r.initialize(32, 32)
    r.set_scene(s)
    P = [scenes.dequantize_positions(g.qpos, g.scaling, g.offset).astype(np.float32) for g in s.geometries[:-1]]
    for force in (False, True):
        r.set_bvh_policy(force_bvh_rebuild=force)
        for gi, p in enumerate(P):
            r.update_vertices(gi, p)
        r.refit()
        got = r.render_ray_queries(q)
        assert np.array_equal(got.view(np.uint32), static_hits.view(np.uint32)), "rebuild" if force else "refit"
    assert r.bvh_rebuild_count() == 2
    osc = O.OracleScene(s)
    for gi, p in enumerate(P):
        osc.set_dynamic_vertices(gi, p)
    assert_ray_visit_parity(r, osc, 48, 32, 1, abi.VARIANT_SIMPLE)
    r.close()


@pytest.mark.parametrize("seed,n_instances", [(31, 9), (32, 24)])
def test_fuzz_rebuild_of_soups(seed, n_instances):
    """the device-side rebuild on what the fuzz scenes throw at it: every mesh dynamic, vertices far from where the host tree was built,
    a triangle collapsed to a point, a mesh flattened to a plane (all Morton codes of one axis equal), sheared / mirrored instances, more
    than 16 of them: brute-force answers, and the oracle walks the exported trees exactly like the device"""
    s = scenes.soup(seed, n_instances=n_instances)
    for m in s.meshes:
        m.dynamic = True
    r = backend.RenderHip()
    r.initialize(64, 64)
    r.set_scene(s)
    r.set_bvh_policy(force_bvh_rebuild=True)
    osc = O.OracleScene(s)
    rng = np.random.default_rng(seed)
    for gi, g in enumerate(s.geometries):
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset)
        P = (P + rng.normal(size=P.shape) * 0.3).astype(np.float32)
        P[0:3] = P[0]
        if gi == 0:
            P[:, 2] = 0.5
        r.update_vertices(gi, P)
        osc.set_dynamic_vertices(gi, P)
    r.refit()
    assert r.bvh_rebuild_count() == len(s.meshes)
    q = random_queries(np.random.default_rng(seed + 1), 20000, -6, 6)
    res = r.render_ray_queries(q)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)), "queries differ from brute force"
    # (sanity: the random queries do meet the scene; the suite's seeds give > 500 hits, a sparse three-instance soup of the soak a few hundred)
    assert (res[:, 0] >= 0).sum() > 100, "only %d of the random queries hit anything" % int((res[:, 0] >= 0).sum())
    assert_ray_visit_parity(r, osc, 64, 64, 1, abi.VARIANT_GLTF)
    r.close()


def test_subtly_dynamic_meshes_are_refitted_never_rebuilt():
    """Mesh::SubtlyDynamic (SceneLoaderParams::small_deformation; the reference builds such meshes for fast tracing + updates,
    render_vulkan.cpp:942-952): vertex updates and refits as for Mesh::Dynamic, but the BVH policy leaves the tree alone"""
    s = scenes.grid(60, 40, deform_t=0.0, name="subtle")
    s.meshes[0].dynamic = abi.MESH_SUBTLY_DYNAMIC
    r = backend.RenderHip()
    r.initialize(64, 48)
    r.set_scene(s)
    r.set_bvh_policy(force_bvh_rebuild=True)
    P = scenes.grid_positions(60, 40, 0.4)
    r.update_vertices(0, P)
    r.refit()
    assert r.bvh_rebuild_count() == 0
    q = _grid_queries(8000, 5)
    res = r.render_ray_queries(q)
    osc = O.OracleScene(s)
    osc.set_dynamic_vertices(0, P)
    ref = np.zeros_like(res)
    osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
    assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)) and (res[:, 0] >= 0).sum() > 500
    r.set_bvh_policy(rebuild_triangle_budget=10)
    r.update_vertices(0, P)
    r.refit()
    assert r.bvh_rebuild_count() == 0
    r.close()


@pytest.mark.library_defaults
@pytest.mark.parametrize("fif", [1, 3])
def test_partially_flattened_scene_with_a_dynamic_mesh(fif, monkeypatch):
    """Round 5 ("partial flattening"): a forest whose first tree mesh is dynamic. On the library's defaults the instances
    of the static meshes become ONE world-space tree beside the dynamic mesh's instance records (before: one dynamic mesh sent the whole
    scene down the two-level walk, 1.5 x slower on C4). Against the two-level build of the same scene (RPTR_FLATTEN=0): the same hit ids for
    ray queries -- bit-identical records where the dynamic mesh is hit, t equal up to the rounding of pre-transformed triangles elsewhere --
    before and after the mesh moves (update_vertices + refit, then a device-side rebuild), the oracle's image within the tolerance, and
    frames in flight on per-context scene copies (fif = 3) bit-identical to frames rendered one at a time."""
    s = scenes.forest(n_meshes=3, tris_per_tree=600, n_instances=40, name="partial-forest")
    s.meshes[0].dynamic = True
    dyn_insts = [k for k, i in enumerate(s.instances) if s.pmeshes[i.pmesh].mesh == 0]
    assert 0 < len(dyn_insts) < len(s.instances) - 1
    q = random_queries(np.random.default_rng(9), 40000, -12, 12)
    q[:, 1] = np.abs(q[:, 1]) * 0.2 + 0.1
    g0 = s.geometries[s.meshes[0].first_geometry]
    P0 = scenes.dequantize_positions(g0.qpos, g0.scaling, g0.offset).astype(np.float32)
    P1 = P0.copy()
    P1[:, 1] *= 1.25   # the tree grows
    W, H, spp = 160, 96, 2

    def run(flatten):
        if flatten is not None:
            monkeypatch.setenv("RPTR_FLATTEN", str(flatten))
        else:
            monkeypatch.delenv("RPTR_FLATTEN", raising=False)
        r = backend.RenderHip(frames_in_flight=fif)
        r.initialize(W, H)
        r.set_scene(s)
        recs = np.frombuffer(np.ascontiguousarray(r.export_bvh()[2]).tobytes(), np.int32).reshape(-1, 32)
        out = {"records": len(recs), "flat_flag": bool((recs[:, 15] & 1).any()), "hits": [], "imgs": []}
        cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
        for step, (P, force) in enumerate(((None, False), (P1, False), (P0, True))):
            if P is not None:
                r.set_bvh_policy(force_bvh_rebuild=force)
                r.update_vertices(s.meshes[0].first_geometry, P)
                r.refit()
            out["hits"].append(r.render_ray_queries(q).copy())
            if fif == 1:
                r.render(cfg, spp=spp)
            else:   # three frames in flight on three scene copies; the last one is the image
                for t in [r.render_async(cfg, spp=spp) for _ in range(3)]:
                    r.wait(t)
            img = np.zeros((H, W, 4), np.float32)
            r.readback_framebuffer(img)
            out["imgs"].append(img)
        out["rebuilds"] = r.bvh_rebuild_count()
        r.close()
        return out

    part, two = run(None), run(0)
    assert part["records"] == 1 + len(dyn_insts) + len(s.instances) and not part["flat_flag"] and not two["flat_flag"]
    assert part["rebuilds"] >= 1 and two["rebuilds"] >= 1
    for a, b in zip(part["hits"], two["hits"]):
        ids_a, ids_b = a[:, 2:].view(np.int32), b[:, 2:].view(np.int32)
        same = (ids_a == ids_b).all(axis=1)
        assert same.mean() > 0.999 and (ids_b[:, 1] >= 0).sum() > 1000
        hit = same & (ids_b[:, 1] >= 0)
        # barycentrics: found on pre-transformed triangles in one build, in object space in the other -- a centimetre-sized triangle ten units
        # from the origin moves by 1e-5 of its size per ulp of a coordinate
        db = np.abs(a[hit, :2] - b[hit, :2])
        assert db.max() < 2e-2 and np.quantile(db, 0.99) < 2e-3, (float(db.max()), float(np.quantile(db, 0.99)))
    for a, b in zip(part["imgs"], two["imgs"]):
        rmse, _, _ = image_error(a, b)
        assert rmse < RMSE_TOL
    if fif == 1:   # and against the oracle, after the last move (P0 again: the scene as loaded)
        osc = O.OracleScene(s)
        ref, _ = osc.render(W, H, spp, variant=abi.VARIANT_GLTF, frame_offset=2 * spp)
        rmse, _, _ = image_error(part["imgs"][2], ref)
        assert rmse < RMSE_TOL
