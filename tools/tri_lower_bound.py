#!/usr/bin/env python3
"""(CPU) A lower bound on the triangle tests per ray of ANY tree over whole-triangle boxes of the C4 forest: for a sample of camera rays, the
number of triangle bounding boxes the ray pierces before its hit -- what a perfect hierarchy with one triangle per leaf (and free inner
nodes) would test -- next to what the host-built tree tests for the same rays (profiles/r03_notes.md section 3).
tools/tri_lower_bound.py [rays]   (a few minutes: 10 M boxes per ray in numpy)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
os.environ["RPTR_FLATTEN"] = "1"
import oracle_lib as O  # noqa: E402
from realtimepathtracingresearchframework_amd import backend, scenes  # noqa: E402

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 150
s = scenes.forest()
T = []
for inst in s.instances:
    pm = s.pmeshes[inst.pmesh]
    mesh = s.meshes[pm.mesh]
    for gi in range(mesh.first_geometry, mesh.first_geometry + mesh.num_geometries):
        g = s.geometries[gi]
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset).reshape(-1, 3, 3)
        M = np.asarray(inst.transform, np.float32)
        T.append(P @ M[:, :3].T + M[:, 3])
T = np.concatenate(T)
lo, hi = T.min(1), T.max(1)
osc = O.OracleScene(s)
osc.build_bvh()
W, H = 480, 270
eye, ctr, up = (np.array(s.camera[k], np.float64) for k in ("eye", "center", "up"))
d = ctr - eye
d /= np.linalg.norm(d)
fov = np.radians(s.camera["fov"])
du = np.cross(d, up)
du = du / np.linalg.norm(du) * 2 * np.tan(fov / 2) * W / H
dv = -np.cross(du, d)
dv = dv / np.linalg.norm(dv) * 2 * np.tan(fov / 2)
rng = np.random.default_rng(1)
pix = rng.uniform(0, 1, (n_rays, 2))
dirs = pix[:, :1] * du + pix[:, 1:] * dv + (d - 0.5 * du - 0.5 * dv)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
o = np.repeat(eye[None].astype(np.float32), n_rays, 0)
tuv, ids = osc.trace_ex(o, dirs.astype(np.float32), 0.0, 1e30)
t_hit = np.where(ids[:, 0] >= 0, tuv[:, 0], np.inf)
counts, t0 = [], time.time()
for r in range(n_rays):
    inv = 1.0 / dirs[r]
    t1, t2 = (lo - eye) * inv, (hi - eye) * inv
    tn, tf = np.minimum(t1, t2).max(1), np.maximum(t1, t2).min(1)
    counts.append(int(((tn <= tf) & (tf >= 0) & (tn < t_hit[r])).sum()))
counts = np.array(counts)
hit = np.isfinite(t_hit)
print("%d camera rays, %d hit: triangle boxes pierced before the hit (lower bound of triangle tests): mean %.2f, hits only %.2f, misses only %.2f  [%.0f s]"
      % (n_rays, hit.sum(), counts.mean(), counts[hit].mean(), counts[~hit].mean() if (~hit).any() else 0.0, time.time() - t0))
nodes, tris, insts, _ = backend.build_bvh_host(s)
osc.import_bvh(nodes, tris, insts)
_, _, vis = osc.trace_ex_counts(o, dirs.astype(np.float32), 0.0, 1e30)
print("host-built tree, same rays: %.2f node visits, %.2f triangle tests per ray" % (vis[:, 0].mean(), vis[:, 1].mean()))
