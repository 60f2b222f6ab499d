"""Per-rank frame time of the tile split on ONE GPU: renders rank 0's share for world = 1, 2, 4, 8 (no gather)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.grid_1m()
W, H, spp = 1920, 1080, 4
for world in (1, 2, 4, 8):
    r = backend.RenderHip(rank=0, world_size=world, stripe_rows=32, stream=torch.cuda.current_stream().cuda_stream)
    r.initialize(W, H); r.set_scene(s)
    cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
    for _ in range(3): r.render(cfg, spp=spp)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 30
    gpu = 0.0
    for _ in range(K):
        st = r.render(cfg, spp=spp); gpu += st.raw.render_time_ms
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / K
    print("world %d: rank-0 share %.3f ms/frame wall, %.3f ms GPU (events), ideal %.3f" % (world, dt, gpu / K, 3.0 / world))
    r.close()
