"""tools/ray_length_tail.py [scene]: the distribution of node visits per ray on the device's tree (C2 by default; `forest`: flattened C4),
rays of a 480 x 270 frame at 1 spp logged by the oracle and walked on the exported tree: what bounds a launch from below is its longest ray
(profiles/r05_notes.md section 16)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, backend, scenes
which = sys.argv[1] if len(sys.argv) > 1 else "grid"
s = scenes.forest() if which == "forest" else scenes.grid_1m()
W, H = 480, 270
r = backend.RenderHip(); r.initialize(W, H); r.set_scene(s)
osc = O.OracleScene(s); osc.import_bvh(*r.export_bvh())
_, st, rays = osc.render_logged(W, H, 1, 4_000_000, variant=abi.VARIANT_SIMPLE, bvh_mode=O.BVH_IMPORTED)
o, tmin, d, tmax, anyhit = rays[:, 0:3], rays[:, 3], rays[:, 4:7], rays[:, 7], rays[:, 8] != 0
for name, sel in (("closest-hit", ~anyhit), ("shadow", anyhit)):
    if not sel.any():
        continue
    _, _, v = osc.trace_ex_counts(np.ascontiguousarray(o[sel]), np.ascontiguousarray(d[sel]), np.ascontiguousarray(tmin[sel]), np.ascontiguousarray(tmax[sel]),
                                  any_hit=(name == "shadow"), bvh_mode=O.BVH_IMPORTED)
    n = v[:, 0].astype(np.int64)
    q = np.percentile(n, [50, 90, 99, 99.9, 99.99])
    print("%s %-12s %8d rays: node visits mean %.1f, median %d, 90 %% %d, 99 %% %d, 99.9 %% %d, 99.99 %% %d, max %d; rays beyond 4 x the mean: %.3f %%, they make %.1f %% of all visits" % (
        which, name, n.size, n.mean(), q[0], q[1], q[2], q[3], q[4], n.max(), 100.0 * (n > 4 * n.mean()).mean(), 100.0 * n[n > 4 * n.mean()].sum() / n.sum()))
