"""ONE frame at a time: the stage-launch schedule against the one-launch frame (csrc/kernels.h rp_k_frame).
tools/frame_probe.py [c2|c3|c5|forest] [world] -- prints ms per frame (host wall clock around render(), GPU events) per setting."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from realtimepathtracingresearchframework_amd import abi, backend, scenes

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if which == "c2":
    scene, W, H, spp, variant = scenes.grid_1m(), 1920, 1080, 4, abi.VARIANT_SIMPLE
elif which == "c3":
    os.environ.setdefault("RPTR_FLATTEN", "1")
    scene, W, H, spp, variant = scenes.grid_1m_lights(), 1920, 1080, 8, abi.VARIANT_GLTF
elif which == "forest":
    os.environ.setdefault("RPTR_FLATTEN", "1")
    scene, W, H, spp, variant = scenes.forest(), 1920, 1080, 4, abi.VARIANT_GLTF
else:
    raise SystemExit("unknown workload")
settings = [("stages", None)] + [("stream " + " ".join("%s=%s" % (k[12:], v) for k, v in e.items()), dict(e, MODE="2")) for e in (
    {}, {"RPTR_STREAM_TRACE_BLOCKS": "3"}, {"RPTR_STREAM_SHADE_BLOCKS": "2", "RPTR_STREAM_TRACE_BLOCKS": "3"}, {"RPTR_STREAM_SHADE_BLOCKS": "2"},
    {"RPTR_STREAM_TRACE_BLOCKS": "2", "RPTR_STREAM_SHADE_BLOCKS": "2"})] + [("one launch " + " ".join("%s=%s" % (k[11:], v) for k, v in e.items()), e) for e in (
    {}, {"RPTR_FRAME_K0": "1"}, {"RPTR_FRAME_K0": "2"}, {"RPTR_FRAME_K0": "4"}, {"RPTR_FRAME_K0": "8"}, {"RPTR_FRAME_PUB": "2"}, {"RPTR_FRAME_PUB": "4"},
    {"RPTR_FRAME_BLOCKS_PER_CU": "3"}, {"RPTR_FRAME_BLOCKS_PER_CU": "2"}, {"RPTR_FRAME_LOCAL_THRESHOLD": "65536"}, {"RPTR_FRAME_LOCAL_THRESHOLD": "1048576"})]
if len(sys.argv) > 3:
    settings = [s for s in settings if s[1] is None or any(a in s[0] for a in sys.argv[3].split(","))]
ref = None
for name, env in settings:
    mode = 0 if env is None else int(env.get("MODE", "1"))
    for k, v in (env or {}).items():
        if k != "MODE":
            os.environ[k] = v
    r = backend.RenderHip(rank=0, world_size=world, stripe_rows=8)
    r.initialize(W, H)
    r.set_scene(scene)
    r.set_frame_schedule(mode)
    r.set_stage_timing(0)
    cfg = backend.RenderConfiguration(scene.camera_params(), active_variant=variant, reset_accumulation=True)
    for _ in range(6):
        st = r.render(cfg, spp=spp)
    n = 40
    t0 = time.perf_counter()
    gpu = 0.0
    for _ in range(n):
        st = r.render(cfg, spp=spp)
        gpu += st.render_time
    wall = (time.perf_counter() - t0) / n * 1e3
    img = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img)
    same = "" if ref is None else ("  image %s" % ("identical" if np.array_equal(img.view(np.uint32), ref.view(np.uint32)) else "DIFFERS"))
    if ref is None:
        ref = img
    sched = r.frame_schedule()
    print("%-44s wall %.3f ms  gpu %.3f ms  rays %d  queues %s polls %s%s" % (name, wall, gpu / n, st.raw.rays_closest + st.raw.rays_shadow, sched[2], sched[3], same), flush=True)
    r.close()
    for k in (env or {}):
        if k != "MODE":
            del os.environ[k]
