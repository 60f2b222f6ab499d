"""re-initialise / re-set_scene on one handle: images stay right, device memory does not grow"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from realtimepathtracingresearchframework_amd import scenes, abi, backend
import oracle_lib as O
from common import image_error
r = backend.RenderHip(frames_in_flight=3)
s1, s2 = scenes.textured_test(), scenes.two_level_test()
for it, (W, H, s) in enumerate([(64, 48, s1), (128, 96, s1), (96, 64, s2), (64, 48, s1), (128, 96, s1), (64, 48, s1)]):
    r.initialize(W, H); r.set_scene(s)
    for k in range(3):
        st = r.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=(k == 0)), spp=2)
    img = np.zeros((H, W, 4), np.float32); assert r.readback_framebuffer(img) == W * H * 4
    ref, _ = O.OracleScene(s).render(W, H, 6)
    free, total = torch.cuda.mem_get_info()
    print(W, H, s.name, "rmse %.2e" % image_error(img, ref)[0], "reported %d MiB" % (r.stats().raw.device_bytes_allocated >> 20), "device in use %d MiB" % ((total - free) >> 20))
r.close()
