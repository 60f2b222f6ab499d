"""tools/texture_cost.py [size]: what the textured instantiations cost (GPU). The 1 M-triangle height field of BASELINE.json, glTF BSDF,
1920x1080, 4 spp, one frame at a time: literal materials against the same materials reading base colour and roughness / metallic from
size x size textures with full mip chains (footprint propagation + anisotropic trilinear lookups).
Round 2: 2.60 -> 3.33 ms per frame (+28 %), all of it in the shade launches (0.59 -> 1.23 ms): the height field is seen at grazing angles,
most lookups take the full 12 taps x 2 levels x 4 texels the sampler's anisotropy asks for. Of the +0.66 ms in the shade launches 0.3 ms are the textured instantiation itself (1 x 1 textures: footprint arithmetic, 16 bytes more
path state, 61 spilled registers), the rest the texel fetches of 1024^2 textures. Tried without effect: the sRGB table in LDS, 3 instead
of 4 waves per SIMD for the textured instantiations (168 VGPRs, 24 spilled), power-of-two wrap without integer division (kept)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from realtimepathtracingresearchframework_amd import abi, backend, scenes  # noqa: E402


def chain(level0):
    out, cur = [], level0.astype(np.float32)
    while cur.shape[0] > 1 or cur.shape[1] > 1:
        h, w = max(1, cur.shape[0] // 2), max(1, cur.shape[1] // 2)
        cur = cur[:2 * h, :2 * w].reshape(h, 2, w, 2, 4).mean(axis=(1, 3)) if min(cur.shape[:2]) > 1 else cur[:h, :w]
        out.append(np.clip(np.round(cur), 0, 255).astype(np.uint8))
    return out


def textured(s, size):
    rng = np.random.default_rng(7)
    for m in s.materials[:8]:
        noise = rng.integers(96, 256, (size, size, 1)).astype(np.float32) / 255.0
        col = np.clip(np.array([m.base_color[0], m.base_color[1], m.base_color[2]], np.float32) * 255.0, 0, 255)
        base = np.concatenate([noise * col, np.full((size, size, 1), 255.0, np.float32)], axis=2).astype(np.uint8)
        spec = np.zeros((size, size, 4), np.uint8)
        spec[..., 0], spec[..., 3] = 127, 255
        spec[..., 1] = np.clip(m.roughness * 255.0 * (0.75 + 0.5 * noise[..., 0]), 0, 255)
        spec[..., 2] = np.clip(m.metallic * 255.0, 0, 255)
        tid = len(s.textures)
        s.textures.append(scenes.Texture(rgba=base, srgb=True, mips=chain(base)))
        s.textures.append(scenes.Texture(rgba=spec, srgb=False, mips=chain(spec)))
        abi.set_float_bits(m.base_color, 0, 0x80000000 | tid)
        m.roughness = abi.textured_param(tid + 1, 1)
        m.metallic = abi.textured_param(tid + 1, 2)
    return s


def time_frames(s, frames=12):
    r = backend.RenderHip(options={"stage_timing": 2})
    r.initialize(1920, 1080)
    r.set_scene(s)
    ms = []
    for _ in range(frames):
        st = r.render(backend.RenderConfiguration(s.camera_params(), active_variant=abi.VARIANT_GLTF, reset_accumulation=True), spp=4)
        ms.append((st.render_time, st.raw.extend_time_ms, st.raw.connect_time_ms, st.raw.shade_only_time_ms, st.raw.tail_time_ms))
    r.close()
    return np.median(np.array(ms[2:]), axis=0)


size = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
plain = time_frames(scenes.grid_1m())
tex = time_frames(textured(scenes.grid_1m(), size))
print("glTF, 4 spp, 1080p, one frame at a time: literal materials %.3f ms, %d x %d mip-mapped textures %.3f ms (+%.0f %%)" % (plain[0], size, size, tex[0], 100 * (tex[0] / plain[0] - 1)))
print("  stages (extend, connect, shade, tail) ms: literal %s, textured %s" % (np.round(plain[1:], 3).tolist(), np.round(tex[1:], 3).tolist()))
