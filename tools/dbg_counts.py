"""Per-ray visit counts GPU vs oracle for every ray of a small-forest render: prints the differing rays."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, backend, scenes

s = scenes.forest(n_meshes=4, tris_per_tree=400, n_instances=36, name="forest-small")
W, H, spp = 160, 90, 2
r = backend.RenderHip(); r.initialize(W, H); r.set_scene(s)
osc = O.OracleScene(s); osc.import_bvh(*r.export_bvh())
_, st, rays = osc.render_logged(W, H, spp, 1 << 20, variant=abi.VARIANT_GLTF, bvh_mode=O.BVH_IMPORTED, count=True)
print("rays logged", len(rays), st.rays_closest + st.rays_shadow)
for any_hit in (False, True):
    sel = rays[rays[:, 8] == (1.0 if any_hit else 0.0)]
    q = np.zeros((len(sel), 8), np.float32)
    q[:, 0:3] = sel[:, 0:3]; q[:, 4:7] = sel[:, 4:7]; q[:, 7] = sel[:, 7]
    res, vis = r.trace_counted(q, tmin=sel[:, 3], any_hit=any_hit)
    tuv, ids, rv = osc.trace_ex_counts(sel[:, 0:3], sel[:, 4:7], sel[:, 3], sel[:, 7], any_hit=any_hit)
    bad = np.nonzero((vis != rv).any(axis=1))[0]
    print("any" if any_hit else "closest", len(sel), "rays;", len(bad), "differ; totals gpu", vis.sum(axis=0), "cpu", rv.sum(axis=0))
    np.set_printoptions(precision=9, floatmode="unique")
    for i in bad[:8]:
        print(i, repr(sel[i]), sel[i].view(np.uint32), "gpu", vis[i], "cpu", rv[i], res[i], tuv[i], ids[i])
    np.save("gpurun_out/bad_rays_%d.npy" % any_hit, sel[bad[:64]])
