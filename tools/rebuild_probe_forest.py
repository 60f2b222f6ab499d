"""Tree quality of the device-side rebuild on thin instanced geometry: the forest of C4 with every tree mesh dynamic (two-level tree), host
SAH trees (refit) against device LBVH trees (rebuild): node visits per ray and frame time."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import numpy as np
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.forest()
for m in s.meshes[:-1]:
    m.dynamic = True
r = backend.RenderHip(options={"stage_timing": 2})
r.initialize(1920, 1080); r.set_scene(s)
cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
def run(label, force):
    r.set_bvh_policy(force_bvh_rebuild=force)
    for gi, g in enumerate(s.geometries[:-1]):
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset).astype(np.float32)
        r.update_vertices(gi, P)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.refit(); e1.record(); torch.cuda.synchronize()
    st = r.render(cfg, spp=4, count_traversal=True).raw
    runs = [r.render(cfg, spp=4).raw for _ in range(3)]
    t = min(x.render_time_ms for x in runs)
    b = min(runs, key=lambda x: x.render_time_ms)
    print('   rays closest %d shadow %d hits shaded %d' % (st.rays_closest, st.rays_shadow, st.hits_shaded), flush=True)
    print('   stages: extend %.2f connect %.2f shade %.2f tail %.2f ms, %d stand-alone bounces' % (b.extend_time_ms, b.connect_time_ms, b.shade_only_time_ms, b.tail_time_ms, b.launches_extend), flush=True)
    print("%-22s refit/rebuild %.2f ms | nodes/closest ray %.2f tris/ray %.2f | shadow nodes/ray %.2f | frame %.3f ms" % (label, e0.elapsed_time(e1),
          st.nodes_closest / st.rays_closest, st.tris_closest / st.rays_closest, (st.nodes_visited - st.nodes_closest) / max(1, st.rays_shadow), t))
def tree_stats(label):
    nodes, tris, insts = r.export_bvh()
    n32 = np.ascontiguousarray(nodes).view(np.int32).reshape(-1, 16)
    child = n32[:, 10:14]
    EMPTY = -(2 ** 31) + 2
    used = (child != EMPTY).any(axis=1)
    inner = (child >= 0)
    leaf = (child <= -2) & (child != EMPTY)
    cnt = ((-2 - child) & 7)[leaf]
    print("%-22s nodes in use %d | children per node %.2f | leaf slots %d with 1/2/3/4 triangles: %s | inner children per node %.2f" % (
        label, int(used.sum()), float(((child != EMPTY).sum(axis=1)[used]).mean()), int(leaf.sum()), np.bincount(cnt, minlength=5)[1:5].tolist(),
        float(inner.sum(axis=1)[used].mean())), flush=True)
run("host SAH (refit)", False)
tree_stats("host SAH")
run("device LBVH (rebuild)", True)
tree_stats("device LBVH")
