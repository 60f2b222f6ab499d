"""Host time of one rptr_hip_render_async call (launch sequence of a frame) and of one wait."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.grid(120, 60)
for lvl in (0, 1, 2):
    r = backend.RenderHip(frames_in_flight=3, stream=torch.cuda.current_stream().cuda_stream)
    r.initialize(256, 256); r.set_scene(s); r.set_stage_timing(lvl)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
    for _ in range(5): r.wait(r.render_async(cfg, spp=1))
    ts, tw = 0.0, 0.0
    K = 200
    for _ in range(K):
        t0 = time.perf_counter(); t = r.render_async(cfg, spp=1); t1 = time.perf_counter(); r.wait(t); t2 = time.perf_counter()
        ts += t1 - t0; tw += t2 - t1
    print("stage timing %d: submit %.1f us, wait (tiny frame, includes its GPU time) %.1f us" % (lvl, ts / K * 1e6, tw / K * 1e6))
    r.close()
