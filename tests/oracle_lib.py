"""ctypes access to the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from realtimepathtracingresearchframework_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle.so")


class OrcRenderArgs(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("row_begin", C.c_int32), ("row_end", C.c_int32),
        ("variant", C.c_int32), ("sample_begin", C.c_int32), ("spp", C.c_int32), ("frame_offset", C.c_uint32),
        ("bvh_mode", C.c_int32), ("n_threads", C.c_int32), ("count_traversal", C.c_int32), ("_pad", C.c_int32),
        ("camera", abi.Camera), ("params", abi.RenderParams), ("scene_params", abi.SceneParams), ("lighting", abi.LightSamplingConfig),
    ]


class OrcRenderStats(C.Structure):
    _fields_ = [
        ("rays_closest", C.c_uint64), ("rays_shadow", C.c_uint64), ("hits_shaded", C.c_uint64),
        ("nodes_closest", C.c_uint64), ("tris_closest", C.c_uint64), ("nodes_shadow", C.c_uint64), ("tris_shadow", C.c_uint64),
        ("seconds", C.c_double), ("threads", C.c_int32), ("_pad", C.c_int32),
    ]


BVH_OWN, BVH_BRUTE, BVH_IMPORTED = 0, 1, 2

_lib = None


def build(force=False):
    """(Re)builds liboracle.so when it is missing or older than its sources (make decides)."""
    subprocess.check_call(["make", "-C", ORACLE_DIR] + (["-B"] if force else []) + ["liboracle.so"], stdout=subprocess.DEVNULL)


BASELINE_LIB_PATH = os.path.join(ORACLE_DIR, "libcpu_baseline.so")


def build_baseline(native_dir=None):
    """The CPU baseline of bench.py: the oracle's own sources built -O3 without diagnostics (oracle/Makefile libcpu_baseline.so). With
    native_dir: a copy built -march=native for THIS host into that directory (None when no compiler is at hand); without: the portable
    x86-64-v3 copy that travels with the repository."""
    if native_dir is not None:
        try:
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-B", "libcpu_baseline.so", "BASELINE_MARCH=native", "BASELINE_OUT=" + native_dir],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            return os.path.join(native_dir, "libcpu_baseline.so")
        except (OSError, subprocess.CalledProcessError):
            return None
    subprocess.check_call(["make", "-C", ORACLE_DIR, "libcpu_baseline.so"], stdout=subprocess.DEVNULL)
    return BASELINE_LIB_PATH


def use_library(path=None):
    """Switches the library behind lib() / OracleScene (None: back to oracle/liboracle.so). Scenes made before the switch belong to the old one."""
    global _lib, LIB_PATH
    _lib = None
    LIB_PATH = path or os.path.join(ORACLE_DIR, "liboracle.so")
    return lib()


def lib():
    global _lib
    if _lib is None:
        if LIB_PATH == os.path.join(ORACLE_DIR, "liboracle.so"):
            build()
        L = C.CDLL(LIB_PATH)
        L.orc_scene_create.restype = C.c_void_p
        L.orc_scene_create.argtypes = [C.POINTER(abi.SceneDesc)]
        L.orc_scene_destroy.argtypes = [C.c_void_p]
        L.orc_scene_build_bvh.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_scene_import_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_scene_set_dynamic_vertices.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.orc_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_trace_ex_counts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_set_ray_log.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_set_ray_log.restype = C.c_size_t
        L.orc_trace_counts.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_trace_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(OrcRenderArgs), C.c_void_p, C.POINTER(OrcRenderStats)]
        L.orc_render_aovs.argtypes = [C.c_void_p, C.POINTER(OrcRenderArgs), C.c_void_p, C.POINTER(OrcRenderStats), C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
        L.orc_resolve_u8.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        L.orc_process_samples_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(abi.RenderParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_scene_set_rng_variant.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_pointset_probe.argtypes = [C.c_void_p] + [C.c_uint32] * 6 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_rng_probe.argtypes = [C.c_uint32] * 5 + [C.POINTER(C.c_uint32), C.c_void_p, C.c_int]
        L.orc_texture_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_texture_probe_ex.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_footprint_probe.argtypes = [C.c_void_p] * 5
        L.orc_hw_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """Owns an oracle scene handle for a realtimepathtracingresearchframework_amd.scenes.Scene."""

    def __init__(self, scene):
        self.scene = scene
        self._desc = scene.desc()
        self.h = lib().orc_scene_create(C.byref(self._desc))

    def close(self):
        if self.h:
            lib().orc_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_rng_variant(self, rng_variant, table=None):
        """the render backend option rng_variant + its table (uint32 words laid out as SobolData / BNData)"""
        if table is None:
            rc = lib().orc_scene_set_rng_variant(self.h, int(rng_variant), None, 0)
        else:
            t = np.ascontiguousarray(table, dtype=np.uint32)
            rc = lib().orc_scene_set_rng_variant(self.h, int(rng_variant), _p(t), t.size)
        assert rc == 0, "orc_scene_set_rng_variant failed"

    def pointset_probe(self, sample_index, frame_offset, frame_id, px, py, dimx, set_dim, dims):
        """(values, index): the draws RANDOM_FLOAT1(rng, dims[i]) after GET_RNG + RANDOM_SET_DIM(set_dim) of one pixel sample"""
        d = np.ascontiguousarray(dims, dtype=np.int32)
        out = np.zeros(d.size, dtype=np.float32)
        idx = C.c_uint32()
        lib().orc_pointset_probe(self.h, sample_index, frame_offset, frame_id, px, py, dimx, set_dim, _p(d), d.size, _p(out), C.byref(idx))
        return out, int(idx.value)

    def build_bvh(self):
        c = (C.c_uint64 * 3)()
        lib().orc_scene_build_bvh(self.h, c)
        return tuple(int(x) for x in c)

    def texture_probe(self, tex_id, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32).reshape(-1, 2)
        out = np.zeros((len(uv), 4), dtype=np.float32)
        lib().orc_texture_probe(self.h, int(tex_id), _p(uv), len(uv), _p(out))
        return out

    def texture_grad(self, tex_id, uv, ddx, ddy):
        """textureGrad of the reference's material sampler (anisotropic, trilinear) as the oracle restates it"""
        q = np.ascontiguousarray(np.concatenate([np.reshape(uv, (-1, 2)), np.reshape(ddx, (-1, 2)), np.reshape(ddy, (-1, 2))], axis=1), dtype=np.float32)
        out = np.zeros((len(q), 4), dtype=np.float32)
        lib().orc_texture_probe_ex(self.h, int(tex_id), 0, _p(q), len(q), _p(out))
        return out

    def texture_lod(self, tex_id, uv, lod):
        uv = np.reshape(uv, (-1, 2))
        q = np.zeros((len(uv), 6), np.float32)
        q[:, :2], q[:, 2] = uv, lod
        out = np.zeros((len(q), 4), dtype=np.float32)
        lib().orc_texture_probe_ex(self.h, int(tex_id), 1, _p(q), len(q), _p(out))
        return out

    def set_dynamic_vertices(self, geometry, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32).reshape(-1, 3)
        rc = lib().orc_scene_set_dynamic_vertices(self.h, int(geometry), _p(xyz), xyz.shape[0])
        assert rc == 0, "orc_scene_set_dynamic_vertices(%d) rejected %d vertices" % (geometry, xyz.shape[0])

    def import_bvh(self, nodes, tris, insts):
        self._bvh_keep = (nodes, tris, insts)
        return lib().orc_scene_import_bvh(self.h, _p(nodes), nodes.size * nodes.itemsize // 64, _p(tris), tris.size * tris.itemsize // 48,
                                          _p(insts), insts.size * insts.itemsize // 128)

    def trace_counts(self, queries, bvh_mode=BVH_IMPORTED):
        """(results (n,4), visits (n,2) uint32): per-query node / triangle visits of the canonical traversal order."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 8)
        out = np.zeros((len(q), 4), dtype=np.float32)
        visits = np.zeros((len(q), 2), dtype=np.uint32)
        rc = lib().orc_trace_counts(self.h, bvh_mode, _p(q), len(q), _p(out), _p(visits))
        assert rc == 0
        return out, visits

    def trace(self, queries, bvh_mode=BVH_OWN, count=False, out=None):
        """queries: (n,8) float32 view of RenderRayQuery[n]. Returns (n,4) float32 [, (nodes,tris)]."""
        q = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, 8)
        if out is None:
            out = np.zeros((len(q), 4), dtype=np.float32)
        cnt = np.zeros(2, dtype=np.uint64)
        rc = lib().orc_trace(self.h, bvh_mode, _p(q), len(q), _p(out), _p(cnt) if count else None)
        assert rc == 0
        return (out, (int(cnt[0]), int(cnt[1]))) if count else out

    def trace_ex(self, o, d, tmin, tmax, any_hit=False, bvh_mode=BVH_OWN, count=False):
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        n = len(o)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,)))
        tuv = np.zeros((n, 3), dtype=np.float32)
        ids = np.zeros((n, 3), dtype=np.int32)
        cnt = np.zeros(2, dtype=np.uint64)
        rc = lib().orc_trace_ex(self.h, bvh_mode, 1 if any_hit else 0, _p(o), _p(d), _p(tmin), _p(tmax), n, _p(tuv), _p(ids),
                                _p(cnt) if count else None)
        assert rc == 0
        return (tuv, ids, (int(cnt[0]), int(cnt[1]))) if count else (tuv, ids)

    def trace_ex_counts(self, o, d, tmin, tmax, any_hit=False, bvh_mode=BVH_IMPORTED):
        """like trace_ex, plus per-ray visits (n,2) uint32."""
        o = np.ascontiguousarray(o, dtype=np.float32)
        d = np.ascontiguousarray(d, dtype=np.float32)
        n = len(o)
        tmin = np.ascontiguousarray(np.broadcast_to(np.asarray(tmin, dtype=np.float32), (n,)))
        tmax = np.ascontiguousarray(np.broadcast_to(np.asarray(tmax, dtype=np.float32), (n,)))
        tuv = np.zeros((n, 3), dtype=np.float32)
        ids = np.zeros((n, 3), dtype=np.int32)
        visits = np.zeros((n, 2), dtype=np.uint32)
        rc = lib().orc_trace_ex_counts(self.h, bvh_mode, 1 if any_hit else 0, _p(o), _p(d), _p(tmin), _p(tmax), n, _p(tuv), _p(ids), None, _p(visits))
        assert rc == 0
        return tuv, ids, visits

    def render_logged(self, width, height, spp, max_rays, **kw):
        """single-threaded render that records every ray: returns (accum, stats, rays (n,9) = o,tmin,d,tmax,any)."""
        buf = np.zeros((max_rays, 9), dtype=np.float32)
        lib().orc_set_ray_log(_p(buf), max_rays)
        try:
            accum, st = self.render(width, height, spp, threads=1, **kw)
        finally:
            n = lib().orc_set_ray_log(None, 0)
        return accum, st, buf[:n]

    def render(self, width, height, spp, variant=abi.VARIANT_GLTF, params=None, lighting=None, rows=None, sample_begin=0,
               frame_offset=0, bvh_mode=BVH_OWN, threads=0, count=False, accum=None, camera=None, scene_params=None, aovs=False,
               prev_camera=None):
        """aovs=True: returns (accum, stats, [albedo_roughness, normal_depth, motion_jitter]) with the three float16 (h, w, 4) AOV
        images the first sample of the frame writes (prev_camera: the previous frame's view, default = the same view)"""
        a = OrcRenderArgs()
        a.width, a.height = width, height
        a.row_begin, a.row_end = rows if rows else (0, height)
        a.variant = variant
        a.sample_begin, a.spp, a.frame_offset = sample_begin, spp, frame_offset
        a.bvh_mode, a.n_threads, a.count_traversal = bvh_mode, threads, 1 if count else 0
        a.camera = camera or self.scene.camera_params()
        a.params = params or abi.RenderParams.default()
        a.scene_params = scene_params or self.scene.scene_params()
        a.lighting = lighting or abi.LightSamplingConfig.default()
        if accum is None:
            accum = np.zeros((height, width, 4), dtype=np.float32)
        st = OrcRenderStats()
        if aovs:
            imgs = [np.zeros((height, width, 4), dtype=np.float16) for _ in range(3)]
            rc = lib().orc_render_aovs(self.h, C.byref(a), _p(accum), C.byref(st), C.byref(prev_camera) if prev_camera is not None else None,
                                       _p(imgs[0]), _p(imgs[1]), _p(imgs[2]))
            assert rc == 0, "orc_render_aovs failed: %d" % rc
            return accum, st, imgs
        rc = lib().orc_render(self.h, C.byref(a), _p(accum), C.byref(st))
        assert rc == 0, "orc_render failed: %d" % rc
        return accum, st


def resolve_u8(accum, exposure=0.0):
    out = np.zeros(accum.shape[:2] + (4,), dtype=np.uint8)
    lib().orc_resolve_u8(_p(np.ascontiguousarray(accum)), accum.shape[0] * accum.shape[1], exposure, _p(out))
    return out


def process_samples_u8(accum, params=None, cam_pos=(0.0, 0.0, 0.0), aovs=None):
    """vulkan/process_samples.comp:134-198 on a resolved RGBA32F image: the RGBA8 frame buffer the reference would show for it
    (exposure, early tone mapping, AOV views of `aovs` = [albedo_roughness, normal_depth, motion_jitter] float16 images, sRGB,
    2x2 replication for render_upscale_factor 2)"""
    params = params or abi.RenderParams.default()
    h, w = accum.shape[:2]
    up = 2 if params.render_upscale_factor == 2 else 1
    out = np.zeros((up * h, up * w, 4), dtype=np.uint8)
    cam = np.asarray(cam_pos, dtype=np.float32)
    a = [np.ascontiguousarray(x).view(np.uint16) for x in aovs] if aovs is not None else [None] * 3
    rc = lib().orc_process_samples_u8(_p(np.ascontiguousarray(accum, dtype=np.float32)), w, h, C.byref(params), _p(cam),
                                      *[(_p(x) if x is not None else None) for x in a], _p(out))
    assert rc == 0
    return out


def halton23(i):
    """entry i of the (2, 3) Halton table behind view_params.screen_jitter, as the oracle regenerates it"""
    out = np.zeros(2, np.float32)
    lib().orc_halton23_probe(C.c_uint32(i), _p(out))
    return out


def footprint_probe(ray_dir, dpdx, dpdy, dst_dir):
    """-> (F 2x2 column major, dpdx', dpdy' recovered from F, reflected F)"""
    a = [np.ascontiguousarray(v, dtype=np.float32) for v in (ray_dir, dpdx, dpdy, dst_dir)]
    out = np.zeros(14, np.float32)
    lib().orc_footprint_probe(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), _p(out))
    return out[0:4].reshape(2, 2).T.copy(), out[4:7].copy(), out[7:10].copy(), out[10:14].reshape(2, 2).T.copy()


def rng_probe(index, frame, px, py, dimx, n=8):
    st = C.c_uint32()
    fl = np.zeros(n, dtype=np.float32)
    lib().orc_rng_probe(index, frame, px, py, dimx, C.byref(st), _p(fl), n)
    return st.value, fl
