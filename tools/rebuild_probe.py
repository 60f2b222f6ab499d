"""GPU time of update_vertices + refit vs device-side rebuild for the 1 M-triangle dynamic grid (one frame context), and the node
visits per closest-hit query of the host's SAH tree vs the device's LBVH tree."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import numpy as np
from realtimepathtracingresearchframework_amd import abi, backend, scenes
NX, NZ = 1000, 500
s = scenes.grid(NX, NZ, deform_t=0.0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
r = backend.RenderHip(stream=stream.cuda_stream)
r.initialize(1920, 1080); r.set_scene(s)
bufs = [torch.from_numpy(scenes.grid_positions(NX, NZ, 0.1 * k)).cuda() for k in range(4)]
cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
def timed(label, force):
    r.set_bvh_policy(force_bvh_rebuild=force)
    ms = []
    for k in range(12):
        b = bufs[k % 4]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r.update_vertices_device(0, b.data_ptr(), b.shape[0]); r.refit(); e1.record()
        torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    st = r.render(cfg, spp=4, count_traversal=True).raw
    st2 = r.render(cfg, spp=4).raw
    print("%-28s update+%s %.3f ms (min of 12) | nodes/closest ray %.2f tris/ray %.2f | frame %.3f ms" % (label, "rebuild" if force else "refit", min(ms[2:]),
          st.nodes_closest / st.rays_closest, st.tris_closest / st.rays_closest, st2.render_time_ms))
timed("host SAH tree, refit", False)
timed("device LBVH, rebuild", True)
timed("device LBVH, refit", False)
print("rebuilds", r.bvh_rebuild_count())
