"""Shared helpers for the parity tests."""
import numpy as np

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, backend

RMSE_TOL = 1e-3  # north_star: validation image error < 1e-3 RMSE at fixed seed/spp


def gpu_render(scene, W, H, spp, variant, rank=0, world=1, stripe_rows=32, count=False, params=None, lighting=None, reset=True,
               renderer=None, keep=False):
    r = renderer or backend.RenderHip(rank=rank, world_size=world, stripe_rows=stripe_rows)
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
