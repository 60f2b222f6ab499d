"""Potential of frames in flight: T host threads, each with its own handle + stream, render frames concurrently."""
import sys, os, time, threading
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from realtimepathtracingresearchframework_amd import abi, backend, scenes
s = scenes.grid_1m()
W, H, spp = 1920, 1080, 4
for world in (1, 8):
    for T in (1, 2, 3):
        hs = []
        for t in range(T):
            st = torch.cuda.Stream()
            r = backend.RenderHip(rank=0, world_size=world, stripe_rows=32, stream=st.cuda_stream)
            r.initialize(W, H); r.set_scene(s)
            hs.append((r, st))
        cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
        K = 30
        def work(r):
            for _ in range(K): r.render(cfg, spp=spp)
        for r, _ in hs: r.render(cfg, spp=spp)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(r,)) for r, _ in hs]
        for x in th: x.start()
        for x in th: x.join()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3 / (K * T)
        print("world %d, %d frames in flight: %.3f ms per frame" % (world, T, dt), flush=True)
        for r, _ in hs: r.close()
