"""The frame as ONE launch driven from device-side queues (csrc/kernels.h rp_k_frame, rptr_hip_set_frame_schedule) against the
sequence of stage launches: same device code per path, so every image must be bit-identical and the same rays must be traced --
whatever the number of bounces with global queues, the slots a block takes at a time, the frame size or the scene's kernels
(alpha test, textures, triangle lights, two-level trees, table-driven point sets).

The queue protocol hands path state from one workgroup to another INSIDE a launch (device-coherent accesses, no kernel boundary):
a stale read would show up here as a differing pixel, which is why the big cases run several frames at full occupancy."""
import os

import numpy as np
import pytest

from common import gpu_render
from realtimepathtracingresearchframework_amd import abi, backend, scenes

pytestmark = pytest.mark.gpu


def _frames(scene, W, H, spp, variant, n_frames, one_launch, env=None, frames_in_flight=1, params=None):
    """one_launch: False / 0 stage launches, True / 1 rp_k_frame, 2 the streaming pair (rp_k_stream_trace + rp_k_stream_shade)"""
    env = dict(env or {})
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        r = backend.RenderHip(frames_in_flight=frames_in_flight)
        r.initialize(W, H)
        r.set_scene(scene)
        r.set_frame_schedule(one_launch)
        out = []
        for k in range(n_frames):
            img, st, _ = gpu_render(scene, W, H, spp, variant, reset=(k == 0), renderer=r, params=params)
            out.append((img.copy(), int(st.raw.rays_closest), int(st.raw.rays_shadow), int(st.raw.hits_shaded)))
        sched = r.frame_schedule()
        r.close()
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return out, sched


def _assert_same(ref, got, what):
    for k, (a, b) in enumerate(zip(ref, got)):
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), "%s: frame %d differs in %d pixels" % (
            what, k, int((a[0].view(np.uint32) != b[0].view(np.uint32)).any(axis=2).sum()))
        assert a[1:] == b[1:], (what, k, a[1:], b[1:])


@pytest.mark.parametrize("scene_name,variant", [("cornell32", abi.VARIANT_SIMPLE), ("cornell32", abi.VARIANT_GLTF), ("two_level_test", abi.VARIANT_GLTF),
                                                ("alpha_test", abi.VARIANT_GLTF), ("glass_test", abi.VARIANT_GLTF_TRANSMISSION)])
def test_one_launch_frames_are_bit_identical_to_stage_launches(scene_name, variant):
    s = getattr(scenes, scene_name)()
    W, H, spp = 160, 120, 3
    ref, _ = _frames(s, W, H, spp, variant, 3, False)
    for env in ({"RPTR_FRAME_PUB": "1"}, {"RPTR_FRAME_PUB": "2"}, {"RPTR_FRAME_PUB": "3", "RPTR_FRAME_K0": "2"}, {"RPTR_FRAME_PUB": "4", "RPTR_FRAME_K0": "4"},
                {"RPTR_FRAME_LOCAL_THRESHOLD": "2000"}):
        got, sched = _frames(s, W, H, spp, variant, 3, True, env)
        assert sched[0] and sched[1] >= 1
        _assert_same(ref, got, "%s %r" % (scene_name, env))


def test_one_launch_frame_of_the_benchmark_scene_at_full_size():
    """C2 at its letter (1 M triangles, 1080p, 4 spp, Lambert): 8.3 M paths through ~8100 slots per bounce on every CU, five frames"""
    s = scenes.grid_1m()
    W, H, spp = 1920, 1080, 4
    ref, _ = _frames(s, W, H, spp, abi.VARIANT_SIMPLE, 5, False)
    got, sched = _frames(s, W, H, spp, abi.VARIANT_SIMPLE, 5, True)
    _assert_same(ref, got, "C2")
    assert sched[1] >= 2 and sched[2][0] >= W * H * spp  # bounce 0 holds every (padded) path, bounce 1 was global
    got2, _ = _frames(s, W, H, spp, abi.VARIANT_SIMPLE, 3, True, {"RPTR_FRAME_PUB": "4", "RPTR_FRAME_K0": "1", "RPTR_FRAME_BLOCKS_PER_CU": "2"})
    _assert_same(ref[:3], got2, "C2 pub 4")


def test_one_launch_frame_with_triangle_lights_and_textures():
    """the glTF kernels with binned-RIS light sampling (LDS exchange inside the shade phase's arena) and textured materials"""
    s = scenes.grid(256, 256, with_emitters=True)
    W, H, spp = 640, 360, 4
    ref, _ = _frames(s, W, H, spp, abi.VARIANT_GLTF, 3, False)
    got, _ = _frames(s, W, H, spp, abi.VARIANT_GLTF, 3, True)
    _assert_same(ref, got, "lights")
    t = scenes.textured_test()
    ref, _ = _frames(t, 320, 240, 3, abi.VARIANT_GLTF, 3, False)
    got, _ = _frames(t, 320, 240, 3, abi.VARIANT_GLTF, 3, True)
    _assert_same(ref, got, "textures")


def test_one_launch_frames_in_flight_and_a_stripe_of_a_split():
    """frames in flight (each context owns its queues) and rank 1 of a 3-way split (ragged tile padding in the identity queue)"""
    s = scenes.cornell32()
    W, H, spp = 200, 136, 2
    ref, _ = _frames(s, W, H, spp, abi.VARIANT_GLTF, 4, False, frames_in_flight=3)
    got, _ = _frames(s, W, H, spp, abi.VARIANT_GLTF, 4, True, frames_in_flight=3)
    _assert_same(ref, got, "in flight")
    imgs = []
    for one in (False, True):
        r = backend.RenderHip(rank=1, world_size=3, stripe_rows=8)
        r.initialize(W, H)
        r.set_scene(s)
        r.set_frame_schedule(one)
        img, st, _ = gpu_render(s, W, H, spp, abi.VARIANT_GLTF, renderer=r)
        imgs.append((img.copy(), int(st.raw.rays_closest)))
        r.close()
    assert np.array_equal(imgs[0][0].view(np.uint32), imgs[1][0].view(np.uint32)) and imgs[0][1] == imgs[1][1]


@pytest.mark.parametrize("scene_name,variant", [("cornell32", abi.VARIANT_SIMPLE), ("cornell32", abi.VARIANT_GLTF), ("two_level_test", abi.VARIANT_GLTF),
                                                ("glass_test", abi.VARIANT_GLTF_TRANSMISSION)])
def test_streaming_frames_are_bit_identical_to_stage_launches(scene_name, variant):
    """schedule 2: a tracer kernel (persistent waves over ITEMS: a path's shadow ray, then its continuation ray, in one lane) and a shader
    kernel run side by side for the whole frame and hand each other chunks through memory (kernels.h "stream"); same device code per path,
    same order of additions into a path's radiance: the same bits, frame after frame (the epoch tags of the rings, the seal at the end)"""
    s = getattr(scenes, scene_name)()
    W, H, spp = 200, 120, 3
    ref, _ = _frames(s, W, H, spp, variant, 4, 0)
    got, sched = _frames(s, W, H, spp, variant, 4, 2)
    assert sched[0] == 2 and sched[1] == -1 and sched[2][0] >= W * H * spp and sched[2][3] >= (W * H * spp) // 1024
    _assert_same(ref, got, scene_name)
    got, _ = _frames(s, W, H, spp, variant, 2, 2, {"RPTR_STREAM_TRACE_BLOCKS": "2", "RPTR_STREAM_SHADE_BLOCKS": "2"})
    _assert_same(ref[:2], got, scene_name + " 2+2 blocks")


def test_streaming_frame_of_the_benchmark_scene_with_lights_at_full_size():
    """C3's kernels (glTF + binned-RIS lights) at 1080p, 2 spp: ~6000 chunks through both rings; and a scene with alpha-tested materials,
    which the streaming schedule hands back to the stage launches"""
    s = scenes.grid_1m_lights()
    W, H, spp = 1920, 1080, 2
    ref, _ = _frames(s, W, H, spp, abi.VARIANT_GLTF, 3, 0)
    got, sched = _frames(s, W, H, spp, abi.VARIANT_GLTF, 3, 2)
    _assert_same(ref, got, "C3 kernels")
    assert sched[1] == -1
    a = scenes.alpha_test()
    ref, _ = _frames(a, 160, 120, 2, abi.VARIANT_GLTF, 2, 0)
    got, sched = _frames(a, 160, 120, 2, abi.VARIANT_GLTF, 2, 2)
    _assert_same(ref, got, "alpha")
    assert sched[1] == 0                                               # (ran as stage launches)
