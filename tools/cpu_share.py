"""Where the host thread spends its time in the pipelined loop (emulated rank 0 of N, 3 frames in flight)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.grid_1m()
W, H, spp = 1920, 1080, 4
FIF = int(os.environ.get("FIF", "3"))
for world in (1, 8):
    r = backend.RenderHip(rank=0, world_size=world, stripe_rows=8, stream=torch.cuda.current_stream().cuda_stream, frames_in_flight=FIF)
    r.initialize(W, H); r.set_scene(s); r.set_stage_timing(1)
    cam = s.camera_params()
    q = []
    for _ in range(6):
        q.append(r.render_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True), spp=spp))
        if len(q) >= FIF: r.wait(q.pop(0))
    while q: r.wait(q.pop(0))
    K = 200; ts = tw = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(K):
        a = time.perf_counter()
        q.append(r.render_async(backend.RenderConfiguration(cam, active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True), spp=spp))
        b = time.perf_counter(); ts += b - a
        if len(q) >= FIF:
            r.wait(q.pop(0)); tw += time.perf_counter() - b
    while q: r.wait(q.pop(0))
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("world %d: %.3f ms/frame; host: submit %.3f ms, blocked in wait %.3f ms per frame" % (world, dt / K * 1e3, ts / K * 1e3, tw / K * 1e3))
    r.close()
