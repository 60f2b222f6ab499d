#!/usr/bin/env python3
"""NS-1 (north_star: "regroup rays by material before the BSDF stages"): where it should pay if anywhere -- C3's shading (glTF BSDF + 512
emissive triangles, 1080p, 8 spp) with 48 distinct TEXTURED materials (base colour sRGB + specular / roughness / metallic, 256 x 256 with
mip chains) instead of 8 literal ones, material patches of 4 x 4 quads so that a 1024-path chunk of a late bounce meets dozens of them.
A/B: the ordering by material fused into the shade kernel's LDS compaction (RPTR_REGROUP=1: no launch, no pass over the queue) against
the plain schedule, one frame at a time (stage times) and pipelined.   python tools/regroup_probe.py [slots]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def scene(slots):
    from realtimepathtracingresearchframework_amd import abi, scenes
    s = scenes.grid(1000, 500, with_emitters=True, name="grid-1M-%d-textured-materials" % slots)
    rng = np.random.default_rng(9)
    nx, nz = 1000, 500
    qi, qj = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    slot = (scenes._hash2(qi // 4, qj // 4, 77) * slots).astype(np.int64) % slots
    s.pmeshes[0].tri_material_ids = np.repeat(slot.reshape(-1), 2).astype(np.uint8)
    emitter = s.materials[-1]
    mats = []
    for k in range(slots):
        base = rng.integers(0, 256, (256, 256, 4)).astype(np.uint8)
        base[..., 3] = 255
        spec = rng.integers(0, 256, (256, 256, 4)).astype(np.uint8)
        spec[..., 1] = np.clip(spec[..., 1], 40, 230)
        tb, ts = scenes.Texture(rgba=base, srgb=True), scenes.Texture(rgba=spec, srgb=False)
        s.textures += [tb, ts]
        m = abi.make_material((0.5, 0.5, 0.5), roughness=0.5, metallic=0.0)
        abi.set_float_bits(m.base_color, 0, abi.float_bits(abi.textured_param(2 * k, 0)))
        for field, ch in (("specular", 0), ("roughness", 1), ("metallic", 2)):
            import ctypes as C
            C.cast(C.byref(m, getattr(abi.BaseMaterial, field).offset), C.POINTER(C.c_uint32))[0] = abi.float_bits(abi.textured_param(2 * k + 1, ch))
        mats.append(m)
    s.materials = mats + [emitter]
    s.pmeshes[1].material_offsets = np.array([slots], np.int32)
    # box-filtered mip chains (the product generates none)
    for t in s.textures:
        lv, cur = [], np.asarray(t.rgba).astype(np.float64)
        while cur.shape[0] > 1:
            cur = cur.reshape(cur.shape[0] // 2, 2, cur.shape[1] // 2, 2, 4).mean(axis=(1, 3))
            lv.append(np.clip(np.round(cur), 0, 255).astype(np.uint8))
        t.mips = lv
    s.prepare_lights()
    return s


def run(s, regroup, W=1920, H=1080, spp=8, frames=24):
    from realtimepathtracingresearchframework_amd import abi, backend
    os.environ["RPTR_REGROUP"] = "1" if regroup else "0"
    out = {}
    cam = s.camera_params()
    cfg = backend.RenderConfiguration(cam, active_variant=abi.VARIANT_GLTF, reset_accumulation=True)
    os.environ["RPTR_SIDE_CONNECT"] = "0"
    r = backend.RenderHip(frames_in_flight=1)
    r.initialize(W, H)
    r.set_scene(s)
    r.set_stage_timing(2)
    for _ in range(3):
        r.render(cfg, spp=spp)
    acc = dict(shade=0.0, total=0.0)
    for _ in range(8):
        st = r.render(cfg, spp=spp).raw
        acc["shade"] += st.shade_only_time_ms / 8
        acc["total"] += st.render_time_ms / 8
    img = np.zeros((H, W, 4), np.float32)
    r.readback_framebuffer(img)
    r.close()
    out["one_at_a_time"] = acc
    del os.environ["RPTR_SIDE_CONNECT"]
    r = backend.RenderHip(frames_in_flight=7)
    r.initialize(W, H)
    r.set_scene(s)
    q = []
    for _ in range(8):
        r.wait(r.render_async(cfg, spp=spp))
    t0 = time.perf_counter()
    for k in range(frames):
        q.append(r.render_async(cfg, spp=spp))
        if len(q) >= 7:
            r.wait(q.pop(0))
    while q:
        r.wait(q.pop(0))
    out["pipelined_ms"] = (time.perf_counter() - t0) * 1e3 / frames
    r.close()
    return out, img


if __name__ == "__main__":
    slots = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    s = scene(slots)
    a, ia = run(s, False)
    b, ib = run(s, True)
    print("%d textured materials, 1080p, 8 spp, glTF + area lights" % slots)
    print("  plain      : shade %.3f ms of %.3f ms per frame (one at a time), %.3f ms pipelined" % (a["one_at_a_time"]["shade"], a["one_at_a_time"]["total"], a["pipelined_ms"]))
    print("  regrouped  : shade %.3f ms of %.3f ms per frame (one at a time), %.3f ms pipelined" % (b["one_at_a_time"]["shade"], b["one_at_a_time"]["total"], b["pipelined_ms"]))
    print("  images bit-identical:", bool(np.array_equal(ia.view(np.uint32), ib.view(np.uint32))))
