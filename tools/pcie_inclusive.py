"""tools/pcie_inclusive.py: what a frame costs a host that takes the image to ITS memory every frame (the contract's "PCIe-inclusive rate":
bench.py's `value` has the output resident in HBM, as the reference's swap-chain path has). C2 (1 M triangles, 1080p, 4 spp, diffuse), one frame
at a time: render, then readback_framebuffer into a numpy buffer -- the RGBA32F accumulation buffer (33 MB) or the RGBA8 display image (8 MB)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from realtimepathtracingresearchframework_amd import abi, backend, scenes

W, H, SPP, N = 1920, 1080, 4, 40
s = scenes.grid_1m()
r = backend.RenderHip(device_ordinal=0)
r.initialize(W, H)
r.set_scene(s)
cfg = backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_SIMPLE, reset_accumulation=True)
f32 = np.zeros((H, W, 4), dtype=np.float32)
u8 = np.zeros((H, W, 4), dtype=np.uint8)
rays = 0
for _ in range(3):
    st = r.render(cfg, spp=SPP)
for name, buf in (("no readback", None), ("RGBA8 display image (8.3 MB)", u8), ("RGBA32F accumulation buffer (33.2 MB)", f32)):
    t0 = time.perf_counter()
    rays = 0
    for _ in range(N):
        st = r.render(cfg, spp=SPP)
        rays += int(st.raw.rays_closest) + int(st.raw.rays_shadow)
        if buf is not None:
            assert r.readback_framebuffer(buf) == W * H * 4
    dt = (time.perf_counter() - t0) / N
    print("%-42s %.3f ms per frame (synchronous, one frame at a time)  %.0f Mrays/s" % (name, dt * 1e3, rays / N / dt * 1e-6))
r.close()
