"""Shared helpers for the parity tests."""
import numpy as np

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, backend

RMSE_TOL = 1e-3  # north_star: validation image error < 1e-3 RMSE at fixed seed/spp


def gpu_render(scene, W, H, spp, variant, rank=0, world=1, stripe_rows=32, count=False, params=None, lighting=None, reset=True,
               renderer=None, keep=False, options=None):
    r = renderer or backend.RenderHip(rank=rank, world_size=world, stripe_rows=stripe_rows, options=options)
    if renderer is None:
        r.initialize(W, H)
        r.set_scene(scene)
    if params is not None:
        r.params = params
    if lighting is not None:
        r.lighting_params = lighting
    cfg = backend.RenderConfiguration(scene.camera_params(), active_variant=variant, reset_accumulation=reset)
    st = r.render(cfg, spp=spp, count_traversal=count)
    img = np.zeros((H, W, 4), dtype=np.float32)
    assert r.readback_framebuffer(img) == W * H * 4
    if keep or renderer is not None:
        return img, st, r
    r.close()
    return img, st, None


def image_error(a, b):
    """RMSE over RGB of the pixels finite in both, plus NaN-mask agreement."""
    fa, fb = np.isfinite(a[..., :3]).all(axis=2), np.isfinite(b[..., :3]).all(axis=2)
    both = fa & fb
    d = (a[..., :3] - b[..., :3])[both]
    rmse = float(np.sqrt(np.mean(d.astype(np.float64) ** 2))) if d.size else 0.0
    return rmse, bool(np.array_equal(fa, fb)), float(np.abs(d).max()) if d.size else 0.0


def random_queries(rng, n, lo, hi, t_max=1e20):
    q = np.zeros((n, 8), np.float32)
    q[:, 0:3] = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    q[:, 4:7] = d
    q[:, 7] = t_max
    return q


def assert_ray_visit_parity(r, osc, W, H, spp, variant, max_rays=1 << 21):
    """Every ray of an oracle render (closest-hit AND occlusion rays, with their real intervals) is replayed through the
    device traversal: results and per-ray node / triangle visit counts must equal the oracle walking the exported tree.
    (Counts of a whole GPU render can differ from the oracle's by a few visits: libm differences of an ulp in a sampled
    direction change that ray's walk. Ray by ray on identical inputs the walk is exact.)"""
    osc.import_bvh(*r.export_bvh())
    _, st, rays = osc.render_logged(W, H, spp, max_rays, variant=variant, bvh_mode=O.BVH_IMPORTED, count=True)
    assert len(rays) == st.rays_closest + st.rays_shadow
    total = np.zeros(2, np.uint64)
    for any_hit in (False, True):
        sel = rays[rays[:, 8] == (1.0 if any_hit else 0.0)]
        q = np.zeros((len(sel), 8), np.float32)
        q[:, 0:3], q[:, 4:7], q[:, 7] = sel[:, 0:3], sel[:, 4:7], sel[:, 7]
        res, vis = r.trace_counted(q, tmin=sel[:, 3], any_hit=any_hit)
        tuv, ids, rv = osc.trace_ex_counts(sel[:, 0:3], sel[:, 4:7], sel[:, 3], sel[:, 7], any_hit=any_hit)
        assert np.array_equal(vis, rv), "%d of %d %s rays walk differently" % ((vis != rv).any(axis=1).sum(), len(sel),
                                                                             "occlusion" if any_hit else "closest-hit")
        if any_hit:
            assert np.array_equal(res[:, 0] != 0, ids[:, 0] != 0)
        else:
            hit = ids[:, 0] >= 0
            assert np.array_equal(res[:, 0] >= 0, hit)
            assert np.array_equal(res[hit, 0:2].view(np.uint32), tuv[hit, 1:3].view(np.uint32))
            assert np.array_equal(res[hit, 3].view(np.int32), ids[hit, 2])
        total += vis.sum(axis=0, dtype=np.uint64)
    assert int(total[0]) == st.nodes_closest + st.nodes_shadow and int(total[1]) == st.tris_closest + st.tris_shadow
    return st
