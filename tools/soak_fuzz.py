"""tools/soak_fuzz.py [first_seed] [count]: the body of tests/test_gpu_parity.py::test_fuzz_soups_trace_and_image and of the rebuild /
refit fuzz tests over many more seeds than the suite runs (GPU). Prints the seeds that fail; exit code = their number.
What to expect (120 seeds, round 2): ray queries and per-ray visit counts never differ; about a quarter of these 96 x 64 / 2 spp images
exceed 1e-3 RMSE through one to six pixels -- late bounces whose rays, a few ulps apart between the device's and the host's libm, fall on
either side of a silhouette of the degenerate soup geometry (replaying the oracle's own rays of those frames on the device gives identical
hits and visit counts). One such pixel is 1.1e-3 RMSE at this size; at the sizes of BASELINE.json the images agree to about 1e-6."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from common import RMSE_TOL, assert_ray_visit_parity, gpu_render, image_error, random_queries  # noqa: E402
from realtimepathtracingresearchframework_amd import abi, backend, scenes  # noqa: E402

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
bad = []
for seed in range(first, first + count):
    try:
        s = scenes.soup(seed)
        r = backend.RenderHip(options={"flatten": 0})  # the bit-exact comparisons pin the two-level form, as tests/conftest.py does (a flattened hit differs from the object-space one by rounding)
        r.initialize(96, 64)
        r.set_scene(s)
        q = random_queries(np.random.default_rng(seed), 20000, -5, 5)
        res = r.render_ray_queries(q)
        osc = O.OracleScene(s)
        ref = np.zeros_like(res)
        osc.trace(q, bvh_mode=O.BVH_BRUTE, out=ref)
        assert np.array_equal(res.view(np.uint32), ref.view(np.uint32)), "ray queries"
        assert_ray_visit_parity(r, osc, 96, 64, 1, abi.VARIANT_GLTF)
        for k, variant in enumerate((abi.VARIANT_GLTF, abi.VARIANT_GLTF_TRANSMISSION)):
            img, _, _ = gpu_render(s, 96, 64, 2, variant, renderer=r)   # (a reset moves frame_offset on by the samples accumulated before)
            ref_img, _ = osc.render(96, 64, 2, variant=variant, frame_offset=2 * k)
            rmse, same, _ = image_error(img, ref_img)
            d = np.abs(img[..., :3] - ref_img[..., :3]).max(axis=2)
            # SOAK_MAX_PIXELS=n: an image beyond the RMSE tolerance still counts as explained when no more than n pixels are off (the
            # silhouette flips of late bounces described above; tests/test_gpu_soak.py runs with 8)
            flips = int((d > 1e-3).sum())
            ok = rmse < RMSE_TOL or flips <= int(os.environ.get("SOAK_MAX_PIXELS", "0"))
            assert same and ok, "image variant %d rmse %g, %d pixels differ by more than 1e-3 (max %g)" % (variant, rmse, flips, d.max())
        r.close()
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("seed", seed, "FAILED:", str(e)[:300], flush=True)
print("%d seeds, %d failed: %s" % (count, len(bad), bad))
sys.exit(len(bad))
