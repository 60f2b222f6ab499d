"""Device LBVH against host SAH on ONE tree over all instanced triangles of the C4 forest (what RPTR_FLATTEN builds on the host):
the 10 M world-space triangles as a single dynamic mesh, refit (host tree) vs rebuild (device tree)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import numpy as np
from realtimepathtracingresearchframework_amd import abi, backend, scenes
src = scenes.forest()
parts = []
for inst in src.instances:
    pm = src.pmeshes[inst.pmesh]
    mesh = src.meshes[pm.mesh]
    for g in src.geometries[mesh.first_geometry:mesh.first_geometry + mesh.num_geometries]:
        P = scenes.dequantize_positions(g.qpos, g.scaling, g.offset).astype(np.float32)
        T = np.asarray(inst.transform, np.float32)
        parts.append((P @ T[:, :3].T + T[:, 3]).astype(np.float32))
allv = np.concatenate(parts).reshape(-1, 3, 3)
print("triangles", len(allv), flush=True)
s = scenes.Scene(name="flat-forest")
s.materials = [abi.make_material((0.35, 0.4, 0.2), roughness=0.8)]
m = scenes._add_mesh(s, allv, dynamic=True)
s.pmeshes.append(scenes.ParameterizedMesh(mesh=m, material_offsets=np.array([0], np.int32)))
s.instances.append(scenes.Instance(transform=scenes.IDENTITY.copy(), pmesh=0))
s.camera, s.config, s.sky_key = src.camera, src.config, src.sky_key
s.prepare_lights()
r = backend.RenderHip()
r.initialize(1920, 1080)
t0 = time.time(); r.set_scene(s); print("set_scene (host SAH build) %.1f s" % (time.time() - t0), flush=True)
cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
P = scenes.dequantize_positions(s.geometries[0].qpos, s.geometries[0].scaling, s.geometries[0].offset).astype(np.float32)
def run(label, force):
    r.set_bvh_policy(force_bvh_rebuild=force)
    r.update_vertices(0, P)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r.refit(); e1.record(); torch.cuda.synchronize()
    st = r.render(cfg, spp=4, count_traversal=True).raw
    t = min(r.render(cfg, spp=4).raw.render_time_ms for _ in range(3))
    print("%-22s refit/rebuild %.2f ms | nodes/closest ray %.2f tris/ray %.2f | shadow nodes/ray %.2f | frame %.3f ms" % (label, e0.elapsed_time(e1),
          st.nodes_closest / st.rays_closest, st.tris_closest / st.rays_closest, (st.nodes_visited - st.nodes_closest) / max(1, st.rays_shadow), t), flush=True)
run("host SAH (refit)", False)
run("device LBVH (rebuild)", True)
