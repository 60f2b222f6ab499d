"""ONE frame at a time on C2: ms per frame, host wall clock, for
  base           one frame context, the default schedule (shadow rays of bounce b on a side stream beside bounce b + 1)
  tail=k         the tail kernel takes over at bounce k (option "tail_bounce")
  halves         the frame as two half-frames (samples 0-1 / 2-3) on two frame contexts, submitted together (round 3's sub-frames)
Every variant's final image is compared with base (bit-identical by construction: the same samples in the same accumulation order).
tools/latency_probe.py [c2|c3]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from realtimepathtracingresearchframework_amd import abi, backend, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    scene, W, H, spp, variant = scenes.grid_1m(), 1920, 1080, 4, abi.VARIANT_SIMPLE
else:
    scene, W, H, spp, variant = scenes.grid_1m_lights(), 1920, 1080, 8, abi.VARIANT_GLTF
cam = scene.camera_params()
N = 60
ref = None


def measure(name, fif, parts, options):
    global ref
    r = backend.RenderHip(frames_in_flight=fif, options=options)
    r.initialize(W, H)
    r.set_scene(scene)
    r.set_stage_timing(0)

    def frame():
        tickets = []
        for k in range(parts):
            cfg = backend.RenderConfiguration(cam, active_variant=variant, reset_accumulation=(k == 0))
            tickets.append(r.render_async(cfg, spp=spp // parts))
        st = None
        for t in tickets:
            st = r.wait(t)
        return st
    for _ in range(6):
        frame()
    t0 = time.perf_counter()
    for _ in range(N):
        st = frame()
    wall = (time.perf_counter() - t0) / N * 1e3
    img = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img)
    same = "" if ref is None else ("  image %s" % ("identical" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "DIFFERS"))
    if ref is None:
        ref = img
    print("%-28s %.3f ms per frame  (spp after: %d)%s" % (name, wall, st.spp, same), flush=True)
    r.close()


measure("base", 1, 1, {})
measure("base, no side stream", 1, 1, {"side_connect": 0})
for k in (1, 2, 3, 4):
    measure("tail=%d" % k, 1, 1, {"tail_bounce": k})
for thr in (16384, 262144, 1048576):
    measure("tail_threshold=%d" % thr, 1, 1, {"tail_threshold": thr})
measure("halves", 2, 2, {})
measure("base again", 1, 1, {})
