"""Procedural scenes for the configurations of BASELINE.json and the
host-side scene container that mirrors the reference's `Scene` closely enough
to drive `set_scene` (librender/scene.h:48-108, mesh.h:10-116).

Vertex streams are produced exactly in the reference's storage format:
unrolled (3 vertices per triangle, no index buffer), positions quantised to
21 bit/axis in a u64, normals oct-encoded 16+16 bit, uvs 16+16 bit
(librender/quantize.h:7-42, restated here in float32 numpy arithmetic).
"""
import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import abi
from . import lights as L

f32 = np.float32


# ------------------------------------------------------------------ quantisation (librender/quantize.h)
def quantize_positions(p, extent, base):
    """quantize.h:7-11 for an (n,3) float32 array -> (n,) uint64."""
    p = np.asarray(p, dtype=f32)
    extent = np.asarray(extent, dtype=f32)
    base = np.asarray(base, dtype=f32)
    q = ((p - base).astype(f32) * f32(0x200000)).astype(f32) / extent
    q = q.astype(f32)
    u = np.minimum(q.astype(np.uint32), np.uint32(0x1FFFFF)).astype(np.uint64)
    return u[:, 0] | (u[:, 1] << np.uint64(21)) | (u[:, 2] << np.uint64(42))


def dequantization_scaling(extent):  # quantize.h:13-15
    return (np.asarray(extent, dtype=f32) / f32(0x200000)).astype(f32)


def dequantization_offset(base, extent):  # quantize.h:16-18
    return (np.asarray(base, dtype=f32) + (np.asarray(extent, dtype=f32) * f32(0.5)).astype(f32) / f32(0x200000)).astype(f32)


def dequantize_positions(q, scaling, offset):
    """librender/dequantize.glsl:8-21."""
    q = np.asarray(q, dtype=np.uint64)
    m = np.uint64(0x1FFFFF)
    x = (q & m).astype(f32)
    y = ((q >> np.uint64(21)) & m).astype(f32)
    z = ((q >> np.uint64(42)) & m).astype(f32)
    v = np.stack([x, y, z], axis=1)
    return (v * np.asarray(scaling, dtype=f32) + np.asarray(offset, dtype=f32)).astype(f32)


def quantize_normals(n):
    """quantize.h:21-35 -> (n,) uint32."""
    n = np.asarray(n, dtype=f32)
    nl1 = (np.abs(n[:, 0]) + np.abs(n[:, 1])).astype(f32) + np.abs(n[:, 2])
    pn = (n[:, :2] / nl1[:, None]).astype(f32)
    fold = n[:, 2] <= 0
    sx = np.where(pn[:, 0] >= 0, f32(1), f32(-1))
    sy = np.where(pn[:, 1] >= 0, f32(1), f32(-1))
    fx = ((f32(1) - np.abs(pn[:, 1])).astype(f32) * sx).astype(f32)
    fy = ((f32(1) - np.abs(pn[:, 0])).astype(f32) * sy).astype(f32)
    px = np.where(fold, fx, pn[:, 0]).astype(f32) * f32(0x8000)
    py = np.where(fold, fy, pn[:, 1]).astype(f32) * f32(0x8000)
    ix = np.clip(px.astype(np.int32), -0x7FFF, 0x7FFF)
    iy = np.clip(py.astype(np.int32), -0x7FFF, 0x7FFF)
    ux = (0x8000 + ix).astype(np.uint32)
    uy = (0x8000 + iy).astype(np.uint32)
    return ux | (uy << np.uint32(16))


def quantize_uvs(uv):
    """quantize.h:38-42 with safety_offset = 0 -> (n,) uint32."""
    uv = np.asarray(uv, dtype=f32)
    k = f32(f32(0xFFFF) / f32(8.0))
    sx = (uv[:, 0] * k).astype(f32)
    sy = ((f32(1.0) - uv[:, 1]).astype(f32) * k).astype(f32)
    ux = (f32(0.5) + sx).astype(f32).astype(np.int64).astype(np.uint32) & np.uint32(0xFFFF)
    uy = (f32(0.5) + sy).astype(f32).astype(np.int64).astype(np.uint32) & np.uint32(0xFFFF)
    return ux | (uy << np.uint32(16))


# ------------------------------------------------------------------ scene container
@dataclass
class Geometry:
    qpos: np.ndarray                     # (3*num_tris,) uint64
    qnrm_uv: Optional[np.ndarray]        # (3*num_tris,) uint64 or None
    num_tris: int
    has_normals: bool
    has_uvs: bool
    scaling: np.ndarray
    offset: np.ndarray


@dataclass
class Mesh:
    first_geometry: int
    num_geometries: int
    dynamic: int = 0   # Mesh::flags (librender/mesh.h:44-47): 1 Dynamic, 2 SubtlyDynamic; bools work as before


@dataclass
class ParameterizedMesh:
    mesh: int
    material_offsets: np.ndarray                 # int32 per geometry
    tri_material_ids: Optional[np.ndarray] = None  # uint8 over all triangles of the mesh


@dataclass
class Instance:
    transform: np.ndarray  # (3,4) float32 object->world
    pmesh: int


@dataclass
class SceneConfig:  # librender/render_params.glsl.h:157-162
    bump_scale: float = 1.0
    sun_dir: tuple = (0.0, 1.0, 0.0)
    turbidity: float = 3.0
    albedo: tuple = (0.2, 0.2, 0.2)


@dataclass
class Texture:
    rgba: np.ndarray  # (height, width, 4) uint8, row 0 first
    srgb: bool = False
    mips: Optional[list] = None  # levels 1..n-1, level l being (max(1, height >> l), max(1, width >> l), 4) uint8; None: level 0 only

    def levels(self):
        """all levels, level 0 first, shapes checked"""
        out = [np.ascontiguousarray(self.rgba, dtype=np.uint8)]
        for m in self.mips or []:
            h, w = out[-1].shape[:2]
            m = np.ascontiguousarray(m, dtype=np.uint8)
            assert m.shape == (max(1, h // 2), max(1, w // 2), 4), "mip level %d has shape %s" % (len(out), m.shape)
            out.append(m)
        return out

    def packed(self):
        """the levels back to back in one buffer (RptrTextureDesc.rgba8); the same buffer for every caller while the texture does not
        change, so that descriptors made earlier (the oracle keeps its scene's) stay valid when another one is made. A texture with mip
        levels is copied into that buffer once: assign new arrays to `rgba` / `mips` instead of writing into the old ones."""
        key = (id(self.rgba),) + tuple(id(m) for m in (self.mips or []))
        cache = getattr(self, "_packed", None)
        if cache is None or cache[0] != key:
            lv = self.levels()
            cache = (key, lv[0] if len(lv) == 1 else np.concatenate([l.reshape(-1) for l in lv]), len(lv), lv[0].shape)
            object.__setattr__(self, "_packed", cache)
        return cache[1], cache[2], cache[3]


@dataclass
class Scene:
    name: str
    geometries: List[Geometry] = field(default_factory=list)
    meshes: List[Mesh] = field(default_factory=list)
    pmeshes: List[ParameterizedMesh] = field(default_factory=list)
    instances: List[Instance] = field(default_factory=list)
    materials: List[abi.BaseMaterial] = field(default_factory=list)
    lights: np.ndarray = field(default_factory=lambda: np.zeros((0, 4, 3), dtype=f32))  # binned TriLightData
    textures: List[Texture] = field(default_factory=list)
    camera: dict = field(default_factory=dict)
    config: SceneConfig = field(default_factory=SceneConfig)
    sky_key: str = ""
    _keep: list = field(default_factory=list, repr=False)

    def dump(self, path, render_params: abi.RenderParams = None, lighting: abi.LightSamplingConfig = None):
        """Writes the scene as a flat little-endian file for the C++ host tools (host/scene_dump.hpp documents the
        layout): what a reference-side adapter would hand over from its own `Scene` (stand-in for the .vks loader)."""
        import struct
        rp = render_params or abi.RenderParams.default()
        lc = lighting or abi.LightSamplingConfig.default()
        with open(path, "wb") as f:
            f.write(b"RPSC1\0\0\0")
            f.write(struct.pack("<6I", len(self.geometries), len(self.meshes), len(self.pmeshes), len(self.instances), len(self.materials),
                                len(self.lights)))
            for g in self.geometries:
                f.write(struct.pack("<3I", g.num_tris, 1 if g.has_normals else 0, 1 if g.has_uvs else 0))
                f.write(np.asarray(g.scaling, dtype=f32).tobytes() + np.asarray(g.offset, dtype=f32).tobytes())
                f.write(struct.pack("<I", 0 if g.qnrm_uv is None else 1))
                f.write(np.ascontiguousarray(g.qpos, dtype=np.uint64).tobytes())
                if g.qnrm_uv is not None:
                    f.write(np.ascontiguousarray(g.qnrm_uv, dtype=np.uint64).tobytes())
            for m in self.meshes:
                f.write(struct.pack("<3I", m.first_geometry, m.num_geometries, int(m.dynamic)))
            for pm in self.pmeshes:
                mo = np.ascontiguousarray(pm.material_offsets, dtype=np.int32)
                f.write(struct.pack("<2I", pm.mesh, len(mo)) + mo.tobytes())
                ti = None if pm.tri_material_ids is None else np.ascontiguousarray(pm.tri_material_ids, dtype=np.uint8)
                f.write(struct.pack("<I", 0 if ti is None else len(ti)))
                if ti is not None:
                    f.write(ti.tobytes())
            for inst in self.instances:
                f.write(np.asarray(inst.transform, dtype=f32).reshape(12).tobytes() + struct.pack("<I", inst.pmesh))
            for m in self.materials:
                f.write(bytes(m))
            if len(self.lights):
                f.write(np.ascontiguousarray(self.lights, dtype=f32).tobytes())
            f.write(bytes(self.camera_params()))
            f.write(bytes(self.scene_params()))
            f.write(bytes(rp))
            f.write(bytes(lc))
            f.write(struct.pack("<I", len(self.textures)))  # optional trailing section (absent in files without textures)
            for t in self.textures:   # third word: srgb flag in the low byte, number of mip levels above it (0 = level 0 only)
                lv = t.levels()
                f.write(struct.pack("<3I", lv[0].shape[1], lv[0].shape[0], (1 if t.srgb else 0) | ((len(lv) if len(lv) > 1 else 0) << 8)))
                for px in lv:
                    f.write(px.tobytes())

    def append(self, other: "Scene"):
        """Scene::Scene(fnames, ...) (librender/scene.cpp:50-69): a further scene file is appended -- meshes, parameterized meshes,
        instances, materials, textures behind what is there, indices shifted (texture handles inside the materials too); camera and
        configuration stay this scene's; the emitters are collected and binned again over the whole scene. Twin of host/scene_dump.hpp
        SceneDump::append."""
        import copy
        import ctypes
        g0, m0, p0, mat0, t0 = len(self.geometries), len(self.meshes), len(self.pmeshes), len(self.materials), len(self.textures)
        self.geometries += list(other.geometries)
        self.meshes += [Mesh(m.first_geometry + g0, m.num_geometries, m.dynamic) for m in other.meshes]
        self.pmeshes += [ParameterizedMesh(pm.mesh + m0, np.asarray(pm.material_offsets, np.int32) + np.int32(mat0), pm.tri_material_ids) for pm in other.pmeshes]
        self.instances += [Instance(inst.transform, inst.pmesh + p0) for inst in other.instances]
        for m in other.materials:
            c = copy.deepcopy(m)
            words = ctypes.cast(ctypes.pointer(c), ctypes.POINTER(ctypes.c_uint32))
            for w in (0, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 19):   # float fields that may hold a texture handle (sign bit set)
                if words[w] & 0x80000000:
                    words[w] = (words[w] & 0xE0000000) | (((words[w] & 0x1FFFFFFF) + t0) & 0x1FFFFFFF)
            if c.normal_map >= 0:
                c.normal_map += t0
            self.materials.append(c)
        self.textures += list(other.textures)
        self.prepare_lights()
        return self

    def merge_partition_instances(self):
        """SceneLoaderParams::PerFile::merge_partition_instances (librender/scene.cpp:757-797): runs of consecutive instances with the same
        transform (the partitions of one exported object), each with its own single-material mesh, become one instance whose mesh lists all
        their geometries. Twin of host/scene_dump.hpp SceneDump::merge_partition_instances. Returns the number of instances merged away."""
        geoms = [list(range(m.first_geometry, m.first_geometry + m.num_geometries)) for m in self.meshes]
        kept, cursor, cursor_t = [], None, None
        for inst in self.instances:
            pm = self.pmeshes[inst.pmesh]
            if pm.tri_material_ids is not None:
                cursor = None
                kept.append(inst)
                continue
            t = np.asarray(inst.transform, f32).tobytes()
            cpm = self.pmeshes[cursor.pmesh] if cursor is not None else None
            if cursor is None or t != cursor_t or int(self.meshes[pm.mesh].dynamic) != int(self.meshes[cpm.mesh].dynamic) or pm.mesh == cpm.mesh:
                cursor, cursor_t = inst, t
                kept.append(inst)
                continue
            geoms[cpm.mesh] += geoms[pm.mesh]
            cpm.material_offsets = np.concatenate([np.asarray(cpm.material_offsets, np.int32), np.asarray(pm.material_offsets, np.int32)])
        merged = len(self.instances) - len(kept)
        self.instances = kept
        if merged:
            new = []
            for m, gl in zip(self.meshes, geoms):
                m.first_geometry, m.num_geometries = len(new), len(gl)
                new += [self.geometries[g] for g in gl]
            self.geometries = new
            self.prepare_lights()
        return merged

    def num_tris(self):
        return sum(g.num_tris for g in self.geometries)

    def num_instanced_tris(self):
        n = 0
        for inst in self.instances:
            m = self.meshes[self.pmeshes[inst.pmesh].mesh]
            n += sum(self.geometries[m.first_geometry + j].num_tris for j in range(m.num_geometries))
        return n

    def prepare_lights(self, lighting: abi.LightSamplingConfig = None):
        """≙ RenderBinnedLightsVulkan::update_scene_from_backend (render_binned_lights.cpp:68-87)."""
        lighting = lighting or abi.LightSamplingConfig.default()
        em = L.collect_emitters(self)
        em, _ = L.update_light_sampling(em, lighting.min_perceived_receiver_dist, lighting.min_radiance, lighting.bin_size)
        self.lights = em
        return self

    def desc(self) -> abi.SceneDesc:
        """Builds the flat RptrSceneDesc; arrays stay alive as long as self does."""
        keep = []
        G = (abi.GeometryDesc * max(1, len(self.geometries)))()
        for i, g in enumerate(self.geometries):
            qpos = np.ascontiguousarray(g.qpos, dtype=np.uint64)
            keep.append(qpos)
            G[i].qpos = qpos.ctypes.data
            if g.qnrm_uv is not None:
                qn = np.ascontiguousarray(g.qnrm_uv, dtype=np.uint64)
                keep.append(qn)
                G[i].qnrm_uv = qn.ctypes.data
            else:
                G[i].qnrm_uv = None
            G[i].num_tris = g.num_tris
            G[i].has_normals = 1 if g.has_normals else 0
            G[i].has_uvs = 1 if g.has_uvs else 0
            G[i].quantized_scaling[:] = [float(x) for x in g.scaling]
            G[i].quantized_offset[:] = [float(x) for x in g.offset]
        M = (abi.MeshDesc * max(1, len(self.meshes)))()
        for i, m in enumerate(self.meshes):
            M[i].first_geometry, M[i].num_geometries, M[i].dynamic = m.first_geometry, m.num_geometries, int(m.dynamic)  # Mesh::flags: 1 Dynamic, 2 SubtlyDynamic
        P = (abi.ParameterizedMeshDesc * max(1, len(self.pmeshes)))()
        for i, p in enumerate(self.pmeshes):
            mo = np.ascontiguousarray(p.material_offsets, dtype=np.int32)
            keep.append(mo)
            P[i].mesh = p.mesh
            P[i].material_offsets = mo.ctypes.data
            if p.tri_material_ids is not None:
                ti = np.ascontiguousarray(p.tri_material_ids, dtype=np.uint8)
                keep.append(ti)
                P[i].tri_material_ids = ti.ctypes.data
            else:
                P[i].tri_material_ids = None
        I = (abi.InstanceDesc * max(1, len(self.instances)))()
        for i, inst in enumerate(self.instances):
            I[i].transform[:] = [float(x) for x in np.asarray(inst.transform, dtype=f32).reshape(-1)]
            I[i].parameterized_mesh = inst.pmesh
        MAT = (abi.BaseMaterial * max(1, len(self.materials)))(*self.materials)
        nl = len(self.lights)
        LT = (abi.TriLightData * max(1, nl))()
        if nl:
            C.memmove(LT, np.ascontiguousarray(self.lights, dtype=f32).ctypes.data, nl * 48)
        d = abi.SceneDesc()
        d.geometries, d.num_geometries = G, len(self.geometries)
        d.meshes, d.num_meshes = M, len(self.meshes)
        d.parameterized_meshes, d.num_parameterized_meshes = P, len(self.pmeshes)
        d.instances, d.num_instances = I, len(self.instances)
        d.materials, d.num_materials = MAT, len(self.materials)
        d.lights, d.num_lights = LT, nl
        TX = (abi.TextureDesc * max(1, len(self.textures)))()
        for i, t in enumerate(self.textures):
            px, n_levels, shape0 = t.packed()   # the levels back to back
            assert len(shape0) == 3 and shape0[2] == 4
            keep.append(px)
            TX[i].rgba8 = px.ctypes.data
            TX[i].height, TX[i].width = shape0[0], shape0[1]
            TX[i].srgb = 1 if t.srgb else 0
            TX[i].mip_levels = n_levels if n_levels > 1 else 0
        d.textures, d.num_textures = TX, len(self.textures)
        keep += [G, M, P, I, MAT, LT, TX]
        self._keep = keep
        return d

    def camera_params(self) -> abi.Camera:
        """eye/center/up/fov -> RenderCameraParams like the app does (pos, dir = normalize(center-eye), up)."""
        eye = np.asarray(self.camera["eye"], dtype=f32)
        center = np.asarray(self.camera["center"], dtype=f32)
        up = np.asarray(self.camera["up"], dtype=f32)
        d = (center - eye).astype(f32)
        d = (d * f32(f32(1.0) / np.sqrt(f32(np.dot(d, d))))).astype(f32)
        cam = abi.Camera()
        cam.pos[:] = [float(x) for x in eye]
        cam.dir[:] = [float(x) for x in d]
        cam.up[:] = [float(x) for x in up]
        cam.fovy = float(self.camera["fov"])
        return cam

    def scene_params(self) -> abi.SceneParams:
        """≙ update_config + update_sky_light (render_vulkan.cpp:2954-2959, render_sky.cpp:25-72).

        The Hosek-Wilkie fit needs the model's published data tables (66 KB RGB + 514 KB spectral coefficients,
        rendering/lights/sky_model_arhosek/sky_model_data_*.h), which this package does not carry: with RPTR_SKY_DATA pointing at
        them the fit runs here (sky_fit.py = host/sky_fit.hpp, bit-equal to the reference's code: tests/test_sky_fit.py); without,
        the built-in scenes take the fitted SkyModelParams / sun radiance of their five configurations from package data generated
        with the reference's own code (data/sky_params.json, written by tools/gen_sky_params.py)."""
        if os.environ.get("RPTR_SKY_DATA"):   # the data headers are at hand: fit this scene's own state (any sun / turbidity / albedo)
            from . import sky_fit
            return sky_fit.fit_sky(_sky_tables(os.environ["RPTR_SKY_DATA"]), self.config.sun_dir, self.config.turbidity, self.config.albedo,
                                   len(self.lights), normal_z_scale=1.0 / self.config.bump_scale)
        sky = load_sky_fixture(self.sky_key, has_lights=len(self.lights) > 0)
        sp = abi.SceneParams()
        for i in range(9):
            sp.sky_params.configs[i][:] = sky["configs"][i]
        sp.sky_params.radiances[:] = sky["radiances"]
        sp.sun_dir[:] = sky["sun_dir"]
        sp.sun_cos_angle = sky["sun_cos_angle"]
        sp.sun_radiance[:] = sky["sun_radiance"]
        sp.normal_z_scale = 1.0 / self.config.bump_scale
        return sp


_SKY_CACHE = None
_SKY_TABLES = {}


def _sky_tables(where):
    if where not in _SKY_TABLES:
        from . import sky_fit
        _SKY_TABLES[where] = sky_fit.SkyTables(where)
    return _SKY_TABLES[where]


def sky_fixture_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sky_params.json")


def load_sky_fixture(key, has_lights):
    global _SKY_CACHE
    if _SKY_CACHE is None:
        with open(sky_fixture_path()) as f:
            _SKY_CACHE = json.load(f)
    e = _SKY_CACHE["entries"][key]
    out = dict(e)
    # render_sky.cpp:67-70: .w = 0.5 with triangle lights, 1.0 without
    out["sun_radiance"] = list(e["sun_radiance_lights" if has_lights else "sun_radiance_nolights"])
    return out


# sky configurations used by the synthetic scenes (sun_dir, turbidity, albedo)
SKY_CONFIGS = {
    "default": dict(sun_dir=(0.0, 1.0, 0.0), turbidity=3.0, albedo=(0.2, 0.2, 0.2)),
    "grid": dict(sun_dir=(0.3, 0.8, 0.5), turbidity=3.0, albedo=(0.2, 0.2, 0.2)),
    "low_sun": dict(sun_dir=(0.8, 0.25, 0.3), turbidity=5.0, albedo=(0.3, 0.3, 0.3)),
    "forest": dict(sun_dir=(-0.4, 0.7, 0.3), turbidity=2.5, albedo=(0.15, 0.2, 0.1)),
    "night": dict(sun_dir=(0.2, -0.3, 0.9), turbidity=3.0, albedo=(0.2, 0.2, 0.2)),
}


def _add_mesh(scene: Scene, tris_xyz, normals=None, uvs=None, dynamic=False):
    """tris_xyz: (n,3,3) float32 world/object positions -> one mesh with one geometry."""
    tris_xyz = np.asarray(tris_xyz, dtype=f32)
    flat = tris_xyz.reshape(-1, 3)
    lo = flat.min(axis=0).astype(f32)
    hi = flat.max(axis=0).astype(f32)
    extent = (hi - lo).astype(f32)
    extent = np.where(extent > 0, extent, f32(1e-3)).astype(f32)  # flat meshes: avoid 0 extent
    qpos = quantize_positions(flat, extent, lo)
    qnu = None
    if normals is not None or uvs is not None:
        qn = quantize_normals(np.asarray(normals, dtype=f32).reshape(-1, 3)) if normals is not None else np.zeros(len(flat), np.uint32)
        qu = quantize_uvs(np.asarray(uvs, dtype=f32).reshape(-1, 2)) if uvs is not None else np.zeros(len(flat), np.uint32)
        qnu = qn.astype(np.uint64) | (qu.astype(np.uint64) << np.uint64(32))
    g = Geometry(qpos=qpos, qnrm_uv=qnu, num_tris=len(tris_xyz), has_normals=normals is not None, has_uvs=uvs is not None,
                 scaling=dequantization_scaling(extent), offset=dequantization_offset(lo, extent))
    scene.geometries.append(g)
    scene.meshes.append(Mesh(first_geometry=len(scene.geometries) - 1, num_geometries=1, dynamic=dynamic))
    return len(scene.meshes) - 1


def _add_mesh_segments(scene: Scene, segments):
    """one mesh with one geometry per entry of `segments` ((n,3,3) positions each), all on the quantisation grid of the whole
    mesh (what a multi-segment .vks mesh looks like, ext/libvkr/src/vkr.h:189-214); flat normals and zero uvs are left out"""
    allp = np.concatenate([np.asarray(t, dtype=f32).reshape(-1, 3) for t in segments])
    lo = allp.min(axis=0).astype(f32)
    extent = (allp.max(axis=0) - lo).astype(f32)
    extent = np.where(extent > 0, extent, f32(1e-3)).astype(f32)
    first = len(scene.geometries)
    for t in segments:
        t = np.asarray(t, dtype=f32)
        scene.geometries.append(Geometry(qpos=quantize_positions(t.reshape(-1, 3), extent, lo), qnrm_uv=None, num_tris=len(t), has_normals=False,
                                         has_uvs=False, scaling=dequantization_scaling(extent), offset=dequantization_offset(lo, extent)))
    scene.meshes.append(Mesh(first_geometry=first, num_geometries=len(segments)))
    return len(scene.meshes) - 1


IDENTITY = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]], dtype=f32)


def _quad(a, b, c, d):
    """two triangles (a,b,c), (a,c,d)"""
    return [[a, b, c], [a, c, d]]


# ------------------------------------------------------------------ C1: Cornell box, 32 triangles
def cornell32() -> Scene:
    """unit box [-1,1]^3: 5 walls (10 tris) + short & tall box without
    bottoms (2x10) + ceiling light quad (2) = 32 triangles; emission 15."""
    s = Scene(name="cornell32")
    T = []
    mats = []
    W, R, Gm, Lm = 0, 1, 2, 3
    # floor (y=-1), ceiling (y=1), back (z=-1): white; left (x=-1): red; right (x=1): green. Normals face inward.
    T += _quad((-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)); mats += [W, W]
    T += _quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)); mats += [W, W]
    T += _quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)); mats += [W, W]
    T += _quad((-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)); mats += [R, R]
    T += _quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)); mats += [Gm, Gm]

    def box(cx, cz, half, h, angle):
        ca, sa = np.cos(angle), np.sin(angle)
        base = []
        for dx, dz in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
            x, z = dx * half, dz * half
            base.append((cx + ca * x - sa * z, cz + sa * x + ca * z))
        y0, y1 = -1.0, -1.0 + h
        out = []
        for k in range(4):
            (x0, z0), (x1, z1) = base[k], base[(k + 1) % 4]
            out += _quad((x1, y0, z1), (x0, y0, z0), (x0, y1, z0), (x1, y1, z1))
        out += _quad((base[0][0], y1, base[0][1]), (base[3][0], y1, base[3][1]), (base[2][0], y1, base[2][1]), (base[1][0], y1, base[1][1]))
        return out
    T += box(0.35, 0.3, 0.3, 0.6, 0.3); mats += [W] * 10
    T += box(-0.35, -0.3, 0.3, 1.2, -0.3); mats += [W] * 10
    T += _quad((-0.25, 0.995, -0.25), (0.25, 0.995, -0.25), (0.25, 0.995, 0.25), (-0.25, 0.995, 0.25)); mats += [Lm, Lm]
    assert len(T) == 32
    mesh = _add_mesh(s, np.array(T, dtype=f32))
    s.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=np.array(mats, np.uint8)))
    s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=0))
    s.materials = [
        abi.make_material((0.73, 0.73, 0.73)),
        abi.make_material((0.65, 0.05, 0.05)),
        abi.make_material((0.12, 0.45, 0.15)),
        abi.make_material((1.0, 1.0, 1.0), emission_intensity=15.0),
    ]
    s.camera = dict(eye=(0, 0, 3.4), center=(0, 0, 0), up=(0, 1, 0), fov=40.0)
    s.config = SceneConfig(**{k: v for k, v in SKY_CONFIGS["default"].items()})
    s.sky_key = "default"
    s.prepare_lights()
    return s


def glass_test(clear=False) -> Scene:
    """Cornell box whose tall box is solid glass (BASE_MATERIAL_ONESIDED: refraction, gltf_bsdf.glsl:308-309,591-593), whose short box
    is frosted thin glass (two-sided: the "double reflection" transmission, :310-311,595-596), plus a thin clear pane across the front:
    the scene of RPTR_VARIANT_GLTF_TRANSMISSION. clear=True: near-specular glass (roughness 0.02 .. 0.05) -- paths through it amplify
    a last-bit difference of sin / cos by 1 / alpha, so single pixels of two correct renderers differ visibly."""
    s = cornell32()
    s.name = "glass"
    ids = s.pmeshes[0].tri_material_ids.copy()
    G1, G2, G3 = len(s.materials), len(s.materials) + 1, len(s.materials) + 2
    ids[10:20] = G2       # short box: thin frosted glass
    ids[20:30] = G1       # tall box: solid glass
    solid = abi.make_material((0.95, 0.98, 1.0), roughness=0.05 if clear else 0.25, ior=1.5, flags=abi.BASE_MATERIAL_NOALPHA | abi.BASE_MATERIAL_ONESIDED)
    solid.specular_transmission = 0.95
    solid.clearcoat_gloss = 0.0025 if clear else 0.04
    frosted = abi.make_material((0.9, 0.7, 0.5), roughness=0.35, ior=1.45)
    frosted.specular_transmission = 0.8
    frosted.clearcoat_gloss = 0.09
    pane = abi.make_material((1.0, 1.0, 1.0), roughness=0.02 if clear else 0.15, ior=1.5)
    pane.specular_transmission = 1.0
    pane.clearcoat_gloss = 0.0004 if clear else 0.0225
    s.materials += [solid, frosted, pane]
    # rebuild the mesh with the pane (2 more triangles)
    g = s.geometries[0]
    P = dequantize_positions(g.qpos, g.scaling, g.offset).reshape(-1, 3, 3)
    T = [tuple(map(tuple, t)) for t in P.tolist()] + _quad((-0.6, -0.5, 0.8), (0.2, -0.5, 0.8), (0.2, 0.4, 0.8), (-0.6, 0.4, 0.8))
    ids = np.concatenate([ids, np.array([G3, G3], np.uint8)])
    s2 = Scene(name="glass")
    mesh = _add_mesh(s2, np.array(T, dtype=f32))
    s2.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=ids))
    s2.instances.append(Instance(transform=IDENTITY.copy(), pmesh=0))
    s2.materials = s.materials
    s2.camera = s.camera
    s2.config = s.config
    s2.sky_key = s.sky_key
    s2.prepare_lights()
    return s2


# ------------------------------------------------------------------ value-noise fbm (scene definition, deterministic)
def _hash2(ix, iz, seed):
    h = (ix.astype(np.uint32) * np.uint32(0x9E3779B1)) ^ (iz.astype(np.uint32) * np.uint32(0x85EBCA77)) ^ np.uint32(seed)
    h ^= h >> np.uint32(16)
    h = h * np.uint32(0x85EBCA6B)
    h ^= h >> np.uint32(13)
    h = h * np.uint32(0xC2B2AE35)
    h ^= h >> np.uint32(16)
    return (h >> np.uint32(8)).astype(np.float64) / float(1 << 24)


def fbm(x, z, seed=1234, octaves=4):
    """4 octaves of hashed value noise, base cell 10 units, in float64 (scene definition only)."""
    total = np.zeros_like(x, dtype=np.float64)
    amp, freq = 1.0, 0.1
    for o in range(octaves):
        fx, fz = x * freq, z * freq
        ix, iz = np.floor(fx), np.floor(fz)
        tx, tz = fx - ix, fz - iz
        sx, sz = tx * tx * (3 - 2 * tx), tz * tz * (3 - 2 * tz)
        ix, iz = ix.astype(np.int64), iz.astype(np.int64)
        a = _hash2(ix, iz, seed + o)
        b = _hash2(ix + 1, iz, seed + o)
        c = _hash2(ix, iz + 1, seed + o)
        d = _hash2(ix + 1, iz + 1, seed + o)
        total += amp * ((a * (1 - sx) + b * sx) * (1 - sz) + (c * (1 - sx) + d * sx) * sz - 0.5)
        amp *= 0.5
        freq *= 2.0
    return total


def _heightfield(nx, nz, x0, x1, z0, z1, height_fn):
    """(nx x nz) quads -> (2*nx*nz, 3, 3) tris, per-vertex normals and uvs (unrolled)."""
    xs = np.linspace(x0, x1, nx + 1)
    zs = np.linspace(z0, z1, nz + 1)
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = height_fn(X, Z)
    # normals from central differences of the analytic height function
    e = 1e-3 * max(x1 - x0, z1 - z0) / max(nx, nz) * 10
    dYdx = (height_fn(X + e, Z) - height_fn(X - e, Z)) / (2 * e)
    dYdz = (height_fn(X, Z + e) - height_fn(X, Z - e)) / (2 * e)
    N = np.stack([-dYdx, np.ones_like(dYdx), -dYdz], axis=-1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    P = np.stack([X, Y, Z], axis=-1)
    # representable uv range of quantize_uv: u in [0,8), v in (-7,1] (16 bit over 8 units, quantize.h:38-42)
    UV = np.stack([(X - x0) / (x1 - x0) * 7.5, 1.0 - (Z - z0) / (z1 - z0) * 7.5], axis=-1)

    def tri_attr(A):
        a00, a10, a01, a11 = A[:-1, :-1], A[1:, :-1], A[:-1, 1:], A[1:, 1:]
        t0 = np.stack([a00, a01, a11], axis=2)  # CCW seen from +y
        t1 = np.stack([a00, a11, a10], axis=2)
        return np.stack([t0, t1], axis=2).reshape(-1, 3, A.shape[-1])
    return tri_attr(P).astype(f32), tri_attr(N).astype(f32), tri_attr(UV).astype(f32)


GRID_MATERIALS = [
    # (base_color, roughness, metallic) for the 8 slots of C2/C3
    ((0.70, 0.65, 0.55), 0.9, 0.0), ((0.35, 0.55, 0.25), 0.7, 0.0), ((0.55, 0.45, 0.35), 0.5, 0.0), ((0.80, 0.80, 0.82), 0.3, 1.0),
    ((0.25, 0.35, 0.60), 0.1, 0.0), ((0.90, 0.75, 0.30), 0.2, 1.0), ((0.60, 0.30, 0.25), 0.6, 0.0), ((0.85, 0.85, 0.85), 0.4, 0.0),
]


def grid(nx=1000, nz=500, with_emitters=False, name=None, deform_t=None) -> Scene:
    """nx x nz quad height field (2*nx*nz triangles) over
    [-50,50]x[-25,25], y = 2*fbm(x,z; seed 1234, 4 octaves), 8 material slots by
    hashed 32x32-quad patches. with_emitters adds 256 emissive quads (C3).
    deform_t (C5): y += 0.5*sin(0.4*x + 2*pi*t)."""
    s = Scene(name=name or ("grid%dx%d" % (nx, nz)))

    def h(X, Z):
        y = 2.0 * fbm(X, Z)
        if deform_t is not None:
            y = y + 0.5 * np.sin(0.4 * X + 2 * np.pi * deform_t)
        return y
    P, N, UV = _heightfield(nx, nz, -50.0, 50.0, -25.0, 25.0, h)
    mesh = _add_mesh(s, P, N, UV, dynamic=deform_t is not None)
    qi, qj = np.meshgrid(np.arange(nx), np.arange(nz), indexing="ij")
    slot = (_hash2(qi // 32, qj // 32, 77) * 8).astype(np.int64) % 8
    tri_mat = np.repeat(slot.reshape(-1), 2).astype(np.uint8)
    s.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=tri_mat))
    s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=0))
    s.materials = [abi.make_material(c, roughness=r, metallic=m) for (c, r, m) in GRID_MATERIALS]
    if with_emitters:
        T = []
        for i in range(16):
            for j in range(16):
                cx = -45.0 + 90.0 * (i + 0.5) / 16
                cz = -22.0 + 44.0 * (j + 0.5) / 16
                hs = 0.4
                # facing down (-y): emitters radiate from their front side
                T += _quad((cx - hs, 6.0, cz - hs), (cx + hs, 6.0, cz - hs), (cx + hs, 6.0, cz + hs), (cx - hs, 6.0, cz + hs))
        em = _add_mesh(s, np.array(T, dtype=f32))
        s.materials.append(abi.make_material((1.0, 0.9, 0.7), emission_intensity=20.0))
        s.pmeshes.append(ParameterizedMesh(mesh=em, material_offsets=np.array([len(s.materials) - 1], np.int32)))
        s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=1))
    s.camera = dict(eye=(0, 12, 40), center=(0, 0, 0), up=(0, 1, 0), fov=65.0)
    s.config = SceneConfig(**SKY_CONFIGS["grid"])
    s.sky_key = "grid"
    s.prepare_lights()
    return s


def grid_positions(nx, nz, deform_t):
    """Unrolled float32 positions (6*nx*nz, 3) of grid(nx, nz) at animation time deform_t: what the app
    writes into the dynamic vertex buffer each frame (C5). Same vertex order as grid()'s geometry 0."""
    def h(X, Z):
        return 2.0 * fbm(X, Z) + 0.5 * np.sin(0.4 * X + 2 * np.pi * deform_t)
    P, _, _ = _heightfield(nx, nz, -50.0, 50.0, -25.0, 25.0, h)
    return np.ascontiguousarray(np.asarray(P, dtype=f32).reshape(-1, 3))


def grid_1m():
    """BASELINE.json configs[1]: procedural 1M-triangle mesh."""
    return grid(1000, 500, name="grid-1M")


def grid_1m_lights():
    """BASELINE.json configs[2]: same mesh + 512 emissive triangles."""
    return grid(1000, 500, with_emitters=True, name="grid-1M-lights")


# ------------------------------------------------------------------ small test scenes
def two_level_test(n_inst=12, seed=5) -> Scene:
    """A few small meshes instanced with random rigid transforms + uniform scale (two-level BVH test)."""
    rng = np.random.default_rng(seed)
    s = Scene(name="two_level_test")
    # mesh 0: bumpy patch with normals/uvs; mesh 1: tetra-ish blob without normals
    P, N, UV = _heightfield(12, 12, -1.0, 1.0, -1.0, 1.0, lambda X, Z: 0.3 * np.sin(3 * X) * np.cos(2 * Z))
    m0 = _add_mesh(s, P, N, UV)
    pts = rng.normal(size=(40, 3, 3)).astype(f32) * 0.6
    m1 = _add_mesh(s, pts)
    s.materials = [abi.make_material((0.8, 0.3, 0.3), roughness=0.5), abi.make_material((0.3, 0.8, 0.3), roughness=0.2, metallic=1.0),
                   abi.make_material((0.3, 0.3, 0.8), roughness=0.8), abi.make_material((1, 1, 1), emission_intensity=8.0)]
    s.pmeshes.append(ParameterizedMesh(mesh=m0, material_offsets=np.array([0], np.int32),
                                       tri_material_ids=(np.arange(288) % 3).astype(np.uint8)))
    s.pmeshes.append(ParameterizedMesh(mesh=m1, material_offsets=np.array([2], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m1, material_offsets=np.array([3], np.int32)))
    for i in range(n_inst):
        ang = rng.uniform(0, 2 * np.pi)
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        Rm = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        sc = rng.uniform(0.6, 1.6)
        t = rng.uniform(-4, 4, size=3)
        M = np.concatenate([Rm * sc, t[:, None]], axis=1).astype(f32)
        s.instances.append(Instance(transform=M, pmesh=int(i % 3)))
    s.camera = dict(eye=(0, 2, 12), center=(0, 0, 0), up=(0, 1, 0), fov=50.0)
    s.config = SceneConfig(**SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s


# ------------------------------------------------------------------ C4: instanced forest
def _tree_mesh(seed, target_tris=10000):
    """A procedural tree (recursive branching prisms + leaf quads) with exactly `target_tris` triangles.
    Returns (tris (n,3,3) float32, per-triangle material id 0 = bark, 1 = leaf)."""
    rng = np.random.default_rng(seed)
    tris, mats = [], []

    def prism(p0, p1, r0, r1, sides=5):
        axis = p1 - p0
        axis /= np.linalg.norm(axis)
        ref = np.array([1.0, 0, 0]) if abs(axis[0]) < 0.9 else np.array([0, 1.0, 0])
        u = np.cross(axis, ref)
        u /= np.linalg.norm(u)
        v = np.cross(axis, u)
        ang = np.linspace(0, 2 * np.pi, sides, endpoint=False)
        ring0 = [p0 + r0 * (np.cos(a) * u + np.sin(a) * v) for a in ang]
        ring1 = [p1 + r1 * (np.cos(a) * u + np.sin(a) * v) for a in ang]
        for k in range(sides):
            a, b, c, d = ring0[k], ring0[(k + 1) % sides], ring1[(k + 1) % sides], ring1[k]
            tris.extend([[a, b, c], [a, c, d]])
            mats.extend([0, 0])

    def leaf(p, size):
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        ref = np.array([0, 1.0, 0]) if abs(n[1]) < 0.9 else np.array([1.0, 0, 0])
        u = np.cross(n, ref)
        u /= np.linalg.norm(u)
        v = np.cross(n, u)
        a, b, c, d = p - size * u - size * v, p + size * u - size * v, p + size * u + size * v, p - size * u + size * v
        tris.extend([[a, b, c], [a, c, d]])
        mats.extend([1, 1])

    tips = []

    def grow(p, d, length, radius, depth):
        q = p + d * length
        prism(p, q, radius, radius * 0.7)
        if depth == 0 or len(tris) > target_tris * 0.45:
            tips.append((q, length))
            return
        for _ in range(int(rng.integers(2, 4))):
            nd = d + rng.normal(size=3) * 0.55
            nd[1] = abs(nd[1]) * 0.6 + 0.25
            nd /= np.linalg.norm(nd)
            grow(q, nd, length * rng.uniform(0.62, 0.8), radius * 0.65, depth - 1)
    grow(np.zeros(3), np.array([0, 1.0, 0]), 1.6, 0.16, 7)
    while len(tris) < target_tris:
        tip, length = tips[int(rng.integers(len(tips)))]
        leaf(tip + rng.normal(size=3) * length * 0.6, rng.uniform(0.05, 0.12))
    return np.array(tris[:target_tris], dtype=f32), np.array(mats[:target_tris], dtype=np.uint8)


def forest(n_meshes=10, tris_per_tree=10000, n_instances=1000, name="forest-10M") -> Scene:
    """n_meshes tree meshes x tris_per_tree triangles, n_instances instances on a jittered grid
    (seed 77, random yaw, uniform scale 0.8-1.3 -- the reference's transform model: translation, uniform scale,
    rotation, ext/libvkr/src/vkr.c:1346-1411) + a 2-triangle ground: 10 000 002 instanced triangles by default."""
    s = Scene(name=name)
    s.materials = [abi.make_material((0.35, 0.25, 0.15), roughness=0.8), abi.make_material((0.2, 0.5, 0.15), roughness=0.6),
                   abi.make_material((0.45, 0.4, 0.3), roughness=0.9)]
    for m in range(n_meshes):
        T, mat = _tree_mesh(m + 1, tris_per_tree)
        mesh = _add_mesh(s, T)
        s.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=mat))
    half = 2.2 * np.sqrt(n_instances) / 2
    g = _add_mesh(s, np.array(_quad((-half, 0, -half), (-half, 0, half), (half, 0, half), (half, 0, -half)), dtype=f32))
    s.pmeshes.append(ParameterizedMesh(mesh=g, material_offsets=np.array([2], np.int32)))
    rng = np.random.default_rng(77)
    side = int(np.ceil(np.sqrt(n_instances)))
    for i in range(n_instances):
        gx, gz = i % side, i // side
        x = (gx + 0.5 + rng.uniform(-0.35, 0.35)) / side * 2 * half - half
        z = (gz + 0.5 + rng.uniform(-0.35, 0.35)) / side * 2 * half - half
        yaw, sc = rng.uniform(0, 2 * np.pi), rng.uniform(0.8, 1.3)
        c, sn = np.cos(yaw) * sc, np.sin(yaw) * sc
        M = np.array([[c, 0, sn, x], [0, sc, 0, 0], [-sn, 0, c, z]], dtype=f32)
        s.instances.append(Instance(transform=M, pmesh=i % n_meshes))
    s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=n_meshes))
    s.camera = dict(eye=(0, 6, half * 0.9), center=(0, 2, 0), up=(0, 1, 0), fov=60.0)
    s.config = SceneConfig(**SKY_CONFIGS["forest"])
    s.sky_key = "forest"
    s.prepare_lights()
    return s


# ------------------------------------------------------------------ fuzz scenes (tests)
def soup(seed, n_meshes=3, tris_per_mesh=200, n_instances=9, degenerate=True) -> Scene:
    """Random triangle soups with the awkward cases a builder / intersector has to survive: slivers, zero-area and
    duplicated triangles, coincident vertices, wildly different triangle sizes, axis-aligned flat meshes, instance
    transforms with non-uniform scale, shear and mirroring. Materials are plain diffuse; no emitters."""
    rng = np.random.default_rng(seed)
    s = Scene(name="soup-%d" % seed)
    s.materials = [abi.make_material((0.7, 0.7, 0.7), roughness=0.8), abi.make_material((0.3, 0.6, 0.8), roughness=0.3, metallic=1.0)]
    for m in range(n_meshes):
        n = tris_per_mesh
        c = rng.uniform(-1, 1, (n, 1, 3))
        size = np.exp(rng.uniform(np.log(1e-3), np.log(0.8), (n, 1, 1)))
        T = (c + rng.normal(size=(n, 3, 3)) * size).astype(f32)
        if degenerate:
            T[0, 1] = T[0, 0]                        # two coincident vertices
            T[1] = T[1, 0]                           # a point
            T[2, 2] = (T[2, 0] + T[2, 1]) / 2        # collinear (zero area up to rounding)
            T[3] = T[4]                              # duplicate triangle
            T[5:15, :, m % 3] = T[5, 0, m % 3]       # a patch of axis-aligned (flat) triangles
        if m == n_meshes - 1:
            T[:, :, 1] = 0.25                        # one mesh entirely flat: zero extent on an axis
        mesh = _add_mesh(s, T)
        ids = rng.integers(0, 2, n).astype(np.uint8)
        s.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=ids))
    for i in range(n_instances):
        A = rng.normal(size=(3, 3))
        q, _ = np.linalg.qr(A)
        scale = np.diag(np.exp(rng.uniform(np.log(0.3), np.log(2.5), 3)))          # non-uniform
        shear = np.eye(3)
        shear[0, 1] = rng.uniform(-0.5, 0.5)
        M3 = q @ scale @ shear
        if i % 3 == 2:
            M3[:, 0] = -M3[:, 0]                                                    # mirrored (negative determinant)
        t = rng.uniform(-3, 3, 3)
        M = np.concatenate([M3, t[:, None]], axis=1).astype(f32)
        s.instances.append(Instance(transform=M, pmesh=int(i % n_meshes)))
    s.camera = dict(eye=(0, 1, 9), center=(0, 0, 0), up=(0, 1, 0), fov=55.0)
    s.config = SceneConfig(**SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s


def book(n_pages=65536) -> Scene:
    """A stack of n_pages parallel unit quads along z: a ray that crosses the pages hits all four children of every node it descends
    through, so a 4-wide tree of depth d parks 3 d entries on the traversal stack before the first triangle is tested -- 65536 pages: depth 8,
    25 entries, more than the 20 the device keeps in LDS (csrc/dtraverse.h RP_LDS_STACK): the scene of the stack-spill parity test."""
    s = Scene(name="book-%d" % n_pages)
    s.materials = [abi.make_material((0.7, 0.7, 0.7), roughness=0.8)]
    T = np.zeros((n_pages, 2, 3, 3), np.float64)
    T[:, 0] = [[0, 0, 0], [1, 0, 0], [1, 1, 0]]
    T[:, 1] = [[0, 0, 0], [1, 1, 0], [0, 1, 0]]
    T[..., 2] = (np.arange(n_pages, dtype=np.float64) / n_pages * 2.0 - 1.0)[:, None, None]
    T[..., 0:2] -= 0.5
    T = T.reshape(-1, 3, 3).astype(f32)
    mesh = _add_mesh(s, T)
    s.pmeshes.append(ParameterizedMesh(mesh=mesh, material_offsets=np.array([0], np.int32), tri_material_ids=np.zeros(len(T), np.uint8)))
    s.instances.append(Instance(transform=np.eye(3, 4, dtype=f32), pmesh=0))
    s.camera = dict(eye=(0.2, 0.1, 5), center=(0, 0, 0), up=(0, 1, 0), fov=20.0)
    s.config = SceneConfig(**SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s


# ------------------------------------------------------------------ textured materials (a8 / a9)
def textured_test(nx=24, nz=24) -> Scene:
    """A bumpy patch and a flat quad whose materials read their parameters from textures: sRGB base colour (checker with a
    gradient), a linear "specular / roughness / metallic" texture read per channel, and a tangent-space normal map (sinusoidal
    dimples); plus a small emissive quad whose emission colour is textured and an untextured wall. UVs tile 3x over the patch
    (REPEAT addressing)."""
    s = Scene(name="textured_test")
    rng = np.random.default_rng(3)
    # texture 0: base colour, sRGB, 16x8 (non-square on purpose)
    yy, xx = np.meshgrid(np.arange(8), np.arange(16), indexing="ij")
    chk = ((xx // 2 + yy // 2) % 2).astype(np.float32)
    base = np.stack([60 + 180 * chk, 40 + 12 * xx, 220 - 20 * yy, np.full_like(chk, 255)], axis=2)
    s.textures.append(Texture(rgba=base.astype(np.uint8), srgb=True))
    # texture 1: specular (r), roughness (g), metallic (b), linear, 8x8 noise
    spec = rng.integers(0, 256, (8, 8, 4)).astype(np.uint8)
    spec[..., 1] = np.clip(spec[..., 1], 40, 230)
    spec[..., 2] = np.where(spec[..., 2] > 128, 255, 0)
    spec[..., 3] = 255
    s.textures.append(Texture(rgba=spec, srgb=False))
    # texture 2: normal map, linear, 32x32 dimples
    v, u = np.meshgrid((np.arange(32) + 0.5) / 32, (np.arange(32) + 0.5) / 32, indexing="ij")
    nxm = 0.45 * np.sin(2 * np.pi * 2 * u)
    nym = 0.45 * np.cos(2 * np.pi * 3 * v)
    nzm = np.sqrt(np.maximum(1 - nxm * nxm - nym * nym, 0))
    nm = np.stack([nxm * 0.5 + 0.5, nym * 0.5 + 0.5, nzm, np.ones_like(nzm)], axis=2)
    s.textures.append(Texture(rgba=np.clip(np.round(nm * 255), 0, 255).astype(np.uint8), srgb=False))
    # texture 3: emission colour, sRGB, 2x2
    s.textures.append(Texture(rgba=np.array([[[255, 200, 120, 255], [255, 240, 200, 255]], [[240, 160, 90, 255], [255, 255, 255, 255]]], np.uint8), srgb=True))

    P, N, UV = _heightfield(nx, nz, -2.0, 2.0, -2.0, 2.0, lambda X, Z: 0.25 * np.sin(1.7 * X) * np.cos(1.3 * Z))
    UV = (np.asarray(UV, dtype=f32).reshape(-1, 2) * f32(3.0)).reshape(np.asarray(UV).shape)   # tile the textures 3x
    m0 = _add_mesh(s, P, N, UV)
    flat = np.array(_quad((-2, 0.9, -2), (2, 0.9, -2), (2, 2.5, -2.4), (-2, 2.5, -2.4)), dtype=f32)
    fuv = np.array([[[0, 0], [2, 0], [2, 1]], [[0, 0], [2, 1], [0, 1]]], dtype=f32)
    fn = np.tile(np.array([0, 0.243, 0.970], dtype=f32), (2, 3, 1))
    m1 = _add_mesh(s, flat, fn, fuv)
    em = np.array(_quad((-0.6, 2.2, 0.8), (0.6, 2.2, 0.8), (0.6, 2.2, -0.2), (-0.6, 2.2, -0.2)), dtype=f32)
    euv = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], dtype=f32)
    m2 = _add_mesh(s, em, np.tile(np.array([0, -1, 0], dtype=f32), (2, 3, 1)), euv)

    def textured(normal_map=-1, emission=0.0, base_tex=0, spec_tex=None):
        m = abi.make_material((0.8, 0.8, 0.8), roughness=0.5, emission_intensity=emission)
        abi.set_float_bits(m.base_color, 0, 0x80000000 | base_tex)
        m.normal_map = normal_map
        if spec_tex is not None:
            for fieldname, ch in (("specular", 0), ("roughness", 1), ("metallic", 2)):
                setattr(m, fieldname, abi.textured_param(spec_tex, ch))
        return m
    s.materials = [textured(normal_map=2, spec_tex=1), textured(normal_map=-1), textured(emission=12.0, base_tex=3),
                   abi.make_material((0.7, 0.7, 0.75), roughness=0.6)]
    s.pmeshes.append(ParameterizedMesh(mesh=m0, material_offsets=np.array([0], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m1, material_offsets=np.array([1], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m2, material_offsets=np.array([2], np.int32)))
    for k in range(3):
        s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=k))
    s.camera = dict(eye=(0.3, 1.6, 4.2), center=(0, 0.5, 0), up=(0, 1, 0), fov=50.0)
    s.config = SceneConfig(**SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s


def alpha_test() -> Scene:
    """Alpha-tested geometry (materials without BASE_MATERIAL_NOALPHA; vulkan/pt_megakernel.glsl:153-212): three
    "foliage" screens behind each other whose base colour texture carries cut-outs (alpha 0), solid texels (alpha 1)
    and a band of fractional alphas (the stochastic branch of the test), two of them instances of one parameterized mesh;
    a screen with per-triangle materials of which only one is alpha-tested; an untextured alpha-tested quad (literal
    colour: alpha 1, always accepted); floor, back wall and an area light so that shadow rays cross the screens."""
    s = Scene(name="alpha_test")
    # texture 0: 16x16 RGBA, sRGB colours; alpha: discs cut out, a fractional band, the rest solid
    yy, xx = np.meshgrid(np.arange(16), np.arange(16), indexing="ij")
    r2 = ((xx % 8) - 3.5) ** 2 + ((yy % 8) - 3.5) ** 2
    alpha = np.where(r2 < 6.0, 0, 255).astype(np.int32)
    alpha[6:10, :] = np.clip(16 * xx[6:10, :] + 8, 0, 255)       # a horizontal band of fractional alphas
    col = np.stack([80 + 10 * xx, 200 - 8 * yy, 60 + 5 * (xx + yy)], axis=2)
    s.textures.append(Texture(rgba=np.concatenate([col, alpha[..., None]], axis=2).astype(np.uint8), srgb=True))
    # texture 1: 4x4, mostly transparent with fractional texels (bilinear filtering makes almost every sample fractional)
    t1 = np.zeros((4, 4, 4), np.int32)
    t1[..., 0] = 230
    t1[..., 1] = 120
    t1[..., 2] = 40
    t1[..., 3] = np.array([[0, 90, 255, 40], [200, 0, 130, 255], [255, 60, 0, 180], [20, 255, 110, 0]])
    s.textures.append(Texture(rgba=t1.astype(np.uint8), srgb=True))

    def screen(z, y0=0.0, y1=2.0, x0=-1.5, x1=1.5, tiles=2.0, cells=1):
        T, U = [], []
        for i in range(cells):
            for j in range(cells):
                ax0, ax1 = x0 + (x1 - x0) * i / cells, x0 + (x1 - x0) * (i + 1) / cells
                ay0, ay1 = y0 + (y1 - y0) * j / cells, y0 + (y1 - y0) * (j + 1) / cells
                u0, u1 = tiles * i / cells, tiles * (i + 1) / cells
                v0, v1 = tiles * j / cells, tiles * (j + 1) / cells
                T += _quad((ax0, ay0, z), (ax1, ay0, z), (ax1, ay1, z), (ax0, ay1, z))
                U += [[[u0, v0], [u1, v0], [u1, v1]], [[u0, v0], [u1, v1], [u0, v1]]]
        T = np.array(T, dtype=f32)
        return T, np.tile(np.array([0, 0, 1], dtype=f32), (len(T), 3, 1)), np.array(U, dtype=f32)

    m_screen = _add_mesh(s, *screen(0.0, cells=2))                       # instanced twice (z = 0.9 and z = 0.3 by transform)
    m_third = _add_mesh(s, *screen(-0.4, tiles=1.0))                    # texture 1
    m_mixed = _add_mesh(s, *screen(1.5, y0=0.0, y1=0.8, x0=-1.5, x1=1.5, tiles=3.0, cells=3))  # per-triangle materials
    m_lit = _add_mesh(s, *screen(1.9, y0=0.0, y1=0.5, x0=-0.5, x1=0.5))   # alpha-tested material with a literal colour
    floor = np.array(_quad((-3, 0, -3), (3, 0, -3), (3, 0, 3), (-3, 0, 3)), dtype=f32)
    wall = np.array(_quad((-3, 0, -1.2), (3, 0, -1.2), (3, 3, -1.2), (-3, 3, -1.2)), dtype=f32)
    m_room = _add_mesh_segments(s, [floor, wall])                          # two segments with a material each
    em = np.array(_quad((-0.5, 2.6, 2.4), (0.5, 2.6, 2.4), (0.5, 2.6, 1.6), (-0.5, 2.6, 1.6)), dtype=f32)
    m_light = _add_mesh(s, em)

    def cutout(tex):
        m = abi.make_material((0.8, 0.8, 0.8), roughness=0.7, flags=0)   # no BASE_MATERIAL_NOALPHA
        abi.set_float_bits(m.base_color, 0, 0x80000000 | tex)
        return m
    s.materials = [cutout(0), cutout(1), abi.make_material((0.2, 0.3, 0.8), roughness=0.4),
                   abi.make_material((0.9, 0.5, 0.1), roughness=0.9, flags=0),
                   abi.make_material((0.75, 0.75, 0.75)), abi.make_material((1.0, 1.0, 1.0), emission_intensity=25.0)]
    s.pmeshes.append(ParameterizedMesh(mesh=m_screen, material_offsets=np.array([0], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m_third, material_offsets=np.array([1], np.int32)))
    mixed_ids = np.array([0, 0, 2, 2, 1, 1, 2, 0, 0, 2, 2, 1, 1, 0, 2, 2, 0, 1], np.uint8)   # materials 0 / 1 (cut-outs) and 2 (opaque)
    s.pmeshes.append(ParameterizedMesh(mesh=m_mixed, material_offsets=np.array([0], np.int32), tri_material_ids=mixed_ids))
    s.pmeshes.append(ParameterizedMesh(mesh=m_lit, material_offsets=np.array([3], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m_room, material_offsets=np.array([4, 2], np.int32)))
    s.pmeshes.append(ParameterizedMesh(mesh=m_light, material_offsets=np.array([5], np.int32)))

    def placed(dx, dy, dz, scale=1.0, yaw=0.0):      # rotation about y x uniform scale: what a .vks instance can hold
        c, sn = np.cos(yaw), np.sin(yaw)
        t = np.zeros((3, 4), f32)
        t[:, :3] = (scale * np.array([[c, 0, sn], [0, 1, 0], [-sn, 0, c]])).astype(f32)
        t[:, 3] = (dx, dy, dz)
        return t
    s.instances.append(Instance(transform=placed(0.0, 0.0, 0.9), pmesh=0))
    s.instances.append(Instance(transform=placed(0.35, 0.1, 0.3, scale=0.9, yaw=0.2), pmesh=0))
    for k in range(1, 6):
        s.instances.append(Instance(transform=IDENTITY.copy(), pmesh=k))
    s.camera = dict(eye=(0.4, 1.3, 4.6), center=(0, 0.9, 0), up=(0, 1, 0), fov=45.0)
    s.config = SceneConfig(**SKY_CONFIGS["low_sun"])
    s.sky_key = "low_sun"
    s.prepare_lights()
    return s
