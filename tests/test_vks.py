"""`.vks` / `.vkt` I/O against the reference's own reader: fixtures under tests/golden/vks were written
by realtimepathtracingresearchframework_amd/vks.py and read back by ext/libvkr/src/vkr.c compiled unmodified
(oracle/_ref/libvkr_ref.so; generator tests/golden/gen_vks_fixture.py). When that library is present (build container,
and the GPU box, where the prebuilt .so travels) the comparison is also made live on freshly written scenes."""
import ctypes as C
import json
import os
import shutil

import numpy as np
import pytest

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, scenes, vks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_LIB = os.path.join(ROOT, "oracle", "_ref", "libvkr_ref.so")


def _bits(x):
    return np.ascontiguousarray(np.asarray(x, np.float32)).view(np.uint32)


def _compare_with_dump(path, ref):
    v = vks.read_vks_header(path)
    for key in ("version", "flags", "headerSize", "dataOffset", "numMaterials", "numTriangles", "numMeshes", "numInstances", "numLodGroups",
                "numFrames", "numStaticTransforms", "numAnimatedTransforms"):
        assert v[key] == ref[key], key
    if v["version"] >= 4:
        assert v["animationOffset"] == ref["animationOffset"]
    assert len(v["meshes"]) == len(ref["meshes"])
    for m, r in zip(v["meshes"], ref["meshes"]):
        assert _bits(m["vertexScale"]).tolist() == r["vertexScale"] and _bits(m["vertexOffset"]).tolist() == r["vertexOffset"]
        for key in ("name", "flags", "numSegments", "materialIdBufferBase", "numMaterialsInRange", "numTriangles", "lodGroup", "vertexBufferOffset",
                    "normalUvBufferOffset", "materialIdBufferOffset", "materialIdSize", "indexBufferOffset", "segmentNumTriangles",
                    "segmentMaterialBaseOffsets"):
            assert m[key] == r[key], key
    assert [(i["name"], i["meshId"], i["transformIndex"], i["flags"]) for i in v["instances"]] == [
        (i["name"], i["meshId"], i["transformIndex"], i["flags"]) for i in ref["instances"]]
    assert [(g["numLevelsOfDetail"], g["meshIds"]) for g in v["lodGroups"]] == [(g["numLevelsOfDetail"], g["meshIds"]) for g in ref["lodGroups"]]
    assert v["materialNames"] == [m["name"] for m in ref["materials"]]
    tdir = vks.texture_dir(path)
    for name, r in zip(v["materialNames"], ref["materials"]):
        m = vks._load_material_files(tdir, name)
        for key in ("emissionIntensity", "specularTransmission", "iorEta", "iorK", "translucency"):
            assert int(_bits(m[key]).reshape(-1)[0]) == r[key], (name, key)
        assert _bits(m["emitterBaseColor"]).tolist() == r["emitterBaseColor"]
        for mine, theirs in (("texBaseColor", "texBaseColor"), ("texNormal", "texNormal"), ("texSpecular", "texSpecular")):
            assert (m[mine] is None) == (r[theirs] is None), (name, mine)
            if m[mine] is not None:
                assert m[mine][0].shape[:2] == (r[theirs]["height"], r[theirs]["width"]) and m[mine][1] == r[theirs]["format"]
                assert 1 + len(m[mine][2]) == r[theirs]["numMipLevels"] and r[theirs]["dataOffset"] == 32 + 24 * r[theirs]["numMipLevels"], (name, mine)
    # the transform table as the reference dequantises it
    assert len(ref["transforms"]) == v["numStaticTransforms"]
    for k, r in enumerate(ref["transforms"]):
        mine = vks.dequantize_transform(v["transforms"][24 * k:24 * k + 24])
        assert _bits(mine).reshape(-1).tolist() == r, "transform %d" % k
    return v


@pytest.mark.parametrize("version", [3, 4])
def test_reader_reproduces_the_reference_readers_dump(version):
    path = os.path.join(GOLD, "vks", "alpha_v%d.vks" % version)
    ref = json.load(open(path[:-4] + ".ref.json"))
    v = _compare_with_dump(path, ref)
    assert v["version"] == version and v["numMeshes"] == 6 and v["numInstances"] == 7


def test_quantization_matches_the_reference_vectors():
    """vkr_dequantize_vertices / _normal_uv / vkr_(de)quantize_transform outputs (fixture from the reference's libvkr) pin the
    oracle's dequantisation (rows a6 / a7) and this module's transform codec. libvkr returns .vks space: x mirrored, y and z
    swapped, normals not normalised (vkr.c:1223-1258); the v coordinate is NOT compared: libvkr computes
    8/65535 * (1 - qv) where the shader (dequantize.glsl:43-48) computes 1 - qv * 8/65535."""
    g = json.load(open(os.path.join(GOLD, "vkr_quantization.json")))
    q = np.array(g["vertex_q"], np.uint64)
    scale = np.array(g["vertex_scale"], np.uint32).view(np.float32)
    offset = np.array(g["vertex_offset"], np.uint32).view(np.float32)
    ref = np.array(g["vertex_out"], np.uint32).view(np.float32).reshape(-1, 3)
    mine = np.zeros((len(q), 3), np.float32)
    O.lib().orc_dequantize_positions(q.ctypes.data_as(C.c_void_p), len(q), scale.ctypes.data_as(C.c_void_p), offset.ctypes.data_as(C.c_void_p),
                                     mine.ctypes.data_as(C.c_void_p))
    assert np.array_equal(_bits(-mine[:, 0]), _bits(ref[:, 0]))
    assert np.array_equal(_bits(mine[:, 2]), _bits(ref[:, 1])) and np.array_equal(_bits(mine[:, 1]), _bits(ref[:, 2]))
    assert np.array_equal(scenes.dequantize_positions(q, scale, offset), mine)
    nq = np.array(g["normal_uv_q"], np.uint64)
    rn = np.array(g["normal_out"], np.uint32).view(np.float32).reshape(-1, 3)
    ruv = np.array(g["uv_out"], np.uint32).view(np.float32).reshape(-1, 2)
    nrm = np.zeros((len(nq), 3), np.float32)
    uv = np.zeros((len(nq), 2), np.float32)
    O.lib().orc_dequantize_normal_uv(nq.ctypes.data_as(C.c_void_p), len(nq), nrm.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p))
    flipped = np.stack([-rn[:, 0], rn[:, 2], rn[:, 1]], axis=1).astype(np.float64)      # back into the shader's axes
    ln = np.linalg.norm(flipped, axis=1, keepdims=True)
    ok = ln[:, 0] > 0
    assert ok.sum() > 200 and np.allclose(nrm[ok], (flipped / np.where(ln > 0, ln, 1))[ok], atol=2e-7)
    assert np.array_equal(_bits(uv[:, 0]), _bits(ruv[:, 0]))
    for m_in, packed, out in zip(g["transform_in"], g["transform_packed"], g["transform_out"]):
        m = np.array(m_in, np.uint32).view(np.float32).reshape(4, 3)
        assert list(vks.quantize_transform(m)) == packed
        assert _bits(vks.dequantize_transform(bytes(packed))).reshape(-1).tolist() == out
    # identity, pure translations, half turns, mirrors, extreme scales, a zero matrix: the packed bytes and the round trip of the
    # reference's vkr_quantize_transform / vkr_dequantize_transform
    assert len(g["special_transform_in"]) >= 16
    for m_in, packed, out in zip(g["special_transform_in"], g["special_transform_packed"], g["special_transform_out"]):
        m = np.array(m_in, np.uint32).view(np.float32).reshape(4, 3)
        with np.errstate(all="ignore"):
            got = list(vks.quantize_transform(m))
            back = _bits(vks.dequantize_transform(bytes(packed))).reshape(-1).tolist()
        assert got == packed, (m.tolist(), got, packed)
        want = np.array(out, np.uint32).view(np.float32)
        have = np.array(back, np.uint32).view(np.float32)
        assert np.array_equal(np.isnan(want), np.isnan(have)) and np.array_equal(np.nan_to_num(want).view(np.uint32), np.nan_to_num(have).view(np.uint32))


def test_block_decoders_known_answers():
    """BC1 / BC4 block layouts: hand-assembled blocks"""
    # c0 = pure red (0xF800) > c1 = pure blue (0x001F): four-colour mode; texel t uses palette entry t % 4
    idx = sum((t % 4) << (2 * t) for t in range(16))
    block = bytes([0x00, 0xF8, 0x1F, 0x00]) + idx.to_bytes(4, "little")
    img = vks.decode_texture(block, 4, 4, vks.FMT_BC1_RGB_UNORM)
    assert img[0].tolist() == [[255, 0, 0, 255], [0, 0, 255, 255], [170, 0, 85, 255], [85, 0, 170, 255]]
    # swapped endpoints: three-colour mode, entry 2 = the mean, entry 3 = transparent black (opaque black for the RGB formats)
    block = bytes([0x1F, 0x00, 0x00, 0xF8]) + idx.to_bytes(4, "little")
    a = vks.decode_texture(block, 4, 4, vks.FMT_BC1_RGBA_UNORM)
    assert a[0].tolist() == [[0, 0, 255, 255], [255, 0, 0, 255], [128, 0, 128, 255], [0, 0, 0, 0]]
    assert vks.decode_texture(block, 4, 4, vks.FMT_BC1_RGB_UNORM)[0, 3].tolist() == [0, 0, 0, 255]
    # BC4: a0 = 255 > a1 = 0 -> eight values; texel t uses entry t % 8
    bits = sum((t % 8) << (3 * t) for t in range(16))
    alpha = bytes([255, 0]) + bits.to_bytes(6, "little")
    img = vks.decode_texture(alpha + bytes([0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0]), 4, 4, vks.FMT_BC3_UNORM)
    assert img[..., 3].reshape(-1)[:8].tolist() == [255, 0, 219, 182, 146, 109, 73, 36] and (img[..., 0] == 255).all()
    # a0 <= a1: six values + 0 and 255
    alpha = bytes([10, 60]) + bits.to_bytes(6, "little")
    rg = vks.decode_texture(alpha + alpha, 4, 4, vks.FMT_BC5_UNORM)
    assert rg[..., 0].reshape(-1)[:8].tolist() == [10, 60, 20, 30, 40, 50, 0, 255] and (rg[..., 2] == 0).all() and (rg[..., 3] == 255).all()
    # a 6x5 image pads to 2x2 blocks and crops back
    rng = np.random.default_rng(5)
    src = rng.integers(0, 256, (5, 6, 4)).astype(np.uint8)
    back = vks.decode_texture(vks.encode_bc5(src), 6, 5, vks.FMT_BC5_UNORM)
    assert back.shape == (5, 6, 4) and np.abs(back[..., :2].astype(int) - src[..., :2].astype(int)).max() <= 255 // 14 + 1


def test_block_encoders_round_trip():
    flat = np.tile(np.array([200, 100, 50, 255], np.uint8), (4, 4, 1))
    assert np.array_equal(vks.decode_texture(vks.encode_bc5(flat), 4, 4, vks.FMT_BC5_UNORM)[..., :2], flat[..., :2])       # 8-bit endpoints: exact
    d = vks.decode_texture(vks.encode_bc1(flat), 4, 4, vks.FMT_BC1_RGB_UNORM).astype(int) - flat.astype(int)
    assert np.abs(d[..., 0]).max() <= 4 and np.abs(d[..., 1]).max() <= 2 and np.abs(d[..., 2]).max() <= 4                 # 5:6:5 endpoints
    yy, xx = np.meshgrid(np.arange(8), np.arange(8), indexing="ij")
    ramp = np.stack([xx * 30, 20 + xx * 25, 10 + xx * 15, 40 + yy * 25], axis=2).astype(np.uint8)     # colours along one line per block
    d3 = vks.decode_texture(vks.encode_bc3(ramp), 8, 8, vks.FMT_BC3_UNORM).astype(int) - ramp.astype(int)
    assert np.abs(d3[..., 3]).max() <= 8 and np.abs(d3[..., :3]).max() <= 40


def _same_streams(a, b):
    assert len(a.pmeshes) == len(b.pmeshes)
    for pa, pb in zip(a.pmeshes, b.pmeshes):
        ma, mb = a.meshes[pa.mesh], b.meshes[pb.mesh]
        assert ma.num_geometries == mb.num_geometries
        assert np.array_equal(pa.material_offsets, pb.material_offsets)
        assert (pa.tri_material_ids is None) == (pb.tri_material_ids is None)
        if pa.tri_material_ids is not None:
            assert np.array_equal(pa.tri_material_ids, pb.tri_material_ids)
        for j in range(ma.num_geometries):
            ga, gb = a.geometries[ma.first_geometry + j], b.geometries[mb.first_geometry + j]
            assert np.array_equal(ga.qpos, gb.qpos) and np.array_equal(_bits(ga.scaling), _bits(gb.scaling)) and np.array_equal(_bits(ga.offset), _bits(gb.offset))
            if ga.qnrm_uv is not None and ga.has_normals and ga.has_uvs:
                assert np.array_equal(ga.qnrm_uv, gb.qnrm_uv)


@pytest.mark.parametrize("name", ["alpha_test", "textured_test", "cornell32", "two_level_test"])
@pytest.mark.parametrize("version", [3, 4])
def test_write_then_read_keeps_the_scene(tmp_path, name, version):
    s = getattr(scenes, name)()
    path = str(tmp_path / (name + ".vks"))
    vks.write_vks(path, s, version=version)
    r = vks.read_vks(path)
    _same_streams(s, r)
    assert len(r.instances) == len(s.instances)
    for ia, ib in zip(s.instances, r.instances):
        assert ia.pmesh == ib.pmesh
        scale = max(1.0, float(np.abs(ia.transform).max()))
        assert np.allclose(ia.transform, ib.transform, atol=2e-4 * scale)       # 16-bit quaternion
    assert len(r.materials) == len(s.materials) and len(r.textures) == 3 * len(s.materials)
    for ma, mb in zip(s.materials, r.materials):
        assert (ma.flags & abi.BASE_MATERIAL_NOALPHA) == (mb.flags & abi.BASE_MATERIAL_NOALPHA)
        assert mb.emission_intensity == pytest.approx(ma.emission_intensity) and mb.ior == pytest.approx(ma.ior)
        # roughness / metallic come back as channels 1 / 2 of the material's third texture, base colour as its first
        tid = abi.float_bits(mb.base_color[0]) & 0x1FFFFFFF if abi.float_bits(mb.base_color[0]) & 0x80000000 else None
        if ma.emission_intensity == 0:
            assert tid is not None and r.textures[tid].srgb
        spec = r.textures[abi.float_bits(mb.roughness) & 0x1FFFFFFF]
        if not (abi.float_bits(ma.roughness) & 0x80000000):
            assert abs(int(spec.rgba[0, 0, 1]) - ma.roughness * 255) <= 3 and abs(int(spec.rgba[0, 0, 2]) - ma.metallic * 255) <= 5
    # the same emitters are found again
    assert len(r.lights) == len(s.lights)
    # and the scene renders: the image equals the original's up to what the format cannot hold (5:6:5 colours, the default
    # normal texel, rounded transforms) -- compared in the mean
    r.camera, r.config, r.sky_key = s.camera, s.config, s.sky_key
    a, _ = O.OracleScene(s).render(64, 48, 8)
    b, _ = O.OracleScene(r).render(64, 48, 8)
    fa, fb = a[..., :3][np.isfinite(a[..., :3])], b[..., :3][np.isfinite(b[..., :3])]
    assert abs(float(fa.mean()) - float(fb.mean())) < 0.05 * float(fa.mean()) + 0.01


def test_lod_groups_instance_the_base_level_only(tmp_path):
    """file version 4 LoD groups (vkr.h:261-270): meshes of a group share their instances' placement, the loader keeps the
    instances of a group's first (base) mesh only (scene.cpp:722-736) -- and the reference's reader sees the same groups"""
    s = scenes.two_level_test()
    n_pm = len(s.pmeshes)
    assert n_pm >= 3
    path = str(tmp_path / "lod.vks")
    vks.write_vks(path, s, lod_groups=[[(0, 0.0), (1, 0.5)]])          # parameterized mesh 1 is a coarser level of mesh 0
    v = vks.read_vks_header(path)
    assert v["numLodGroups"] == 2 and v["lodGroups"][1]["meshIds"] == [0, 1] and v["lodGroups"][1]["detailReduction"] == [0.0, 0.5]
    assert [m["lodGroup"] for m in v["meshes"]][:3] == [1, 1, 0]
    r = vks.read_vks(path)
    kept = [i for i in s.instances if i.pmesh != 1]
    assert len(r.instances) == len(kept) and all(a.pmesh == b.pmesh for a, b in zip(r.instances, kept))
    if os.path.isfile(REF_LIB):
        out = str(tmp_path / "dump.json")
        assert _ref().ref_vkr_dump(path.encode(), out.encode()) == 0
        ref = json.load(open(out))
        _compare_with_dump(path, ref)
        assert [g["detailReduction"] for g in ref["lodGroups"]][1] == _bits([0.0, 0.5]).tolist()
    with pytest.raises(vks.VksError):
        vks.write_vks(str(tmp_path / "lod3.vks"), s, version=3, lod_groups=[[(0, 0.0), (1, 0.5)]])


def test_remove_first_lods_swaps_in_the_coarser_level(tmp_path):
    """SceneLoaderParams::PerFile::remove_first_LODs (scene.cpp:801-815) + unlink_pruned_lod_meshes (:229-246): the instances of a LoD
    group render level n instead of the base level (the coarsest one when the group has fewer levels)"""
    s = scenes.two_level_test()
    path = str(tmp_path / "lod.vks")
    vks.write_vks(path, s, lod_groups=[[(0, 0.0), (1, 0.5)]])
    base = vks.read_vks(path)
    lvl1 = vks.read_vks(path, remove_first_lods=1)
    lvl9 = vks.read_vks(path, remove_first_lods=9)
    assert len(lvl1.instances) == len(base.instances)
    assert [i.pmesh for i in lvl1.instances] == [1 if i.pmesh == 0 else i.pmesh for i in base.instances]
    assert [i.pmesh for i in lvl9.instances] == [i.pmesh for i in lvl1.instances]
    assert any(i.pmesh == 0 for i in base.instances) and all(np.array_equal(a.transform, b.transform) for a, b in zip(base.instances, lvl1.instances))


def test_sixteen_bit_material_ids_and_index_buffers(tmp_path):
    """a mesh whose material range exceeds 256 stores two bytes per triangle id (vkr.c:1127-1130); the reference's backend uploads
    static_cast<uint8_t>(id) (render_vulkan.cpp:1114-1126), so does the reader. VKR_MESH_FLAGS_INDICES files carry a 12-byte-per-triangle
    index buffer behind the ids (vkr.c:1131-1136): skipped, the vertex streams are unrolled. libvkr reads the same offsets."""
    s = scenes.alpha_test()
    pm = next(i for i, p in enumerate(s.pmeshes) if p.tri_material_ids is not None)
    n = len(s.pmeshes[pm].tri_material_ids)
    wide = (np.asarray(s.pmeshes[pm].tri_material_ids, np.uint16) + np.uint16(256) * (np.arange(n) % 3).astype(np.uint16)).astype(np.uint16)
    path = str(tmp_path / "wide.vks")
    vks.write_vks(path, s, wide_material_ids={pm: wide}, index_buffers=True)
    v = vks.read_vks_header(path)
    assert v["meshes"][pm]["materialIdSize"] == 2 and v["meshes"][pm]["numMaterialsInRange"] > 0x100
    assert all(m["flags"] & vks.MESH_FLAGS_INDICES and m["indexBufferOffset"] > m["materialIdBufferOffset"] for m in v["meshes"])
    r = vks.read_vks(path)
    assert np.array_equal(r.pmeshes[pm].tri_material_ids, (wide & 0xFF).astype(np.uint8))
    assert np.array_equal(r.pmeshes[pm].tri_material_ids, np.asarray(s.pmeshes[pm].tri_material_ids, np.uint8))
    plain = str(tmp_path / "plain.vks")
    vks.write_vks(plain, s)
    q = vks.read_vks(plain)
    assert all(np.array_equal(a.qpos, b.qpos) for a, b in zip(r.geometries, q.geometries)) and len(r.instances) == len(q.instances)
    if os.path.isfile(REF_LIB):
        out = str(tmp_path / "dump.json")
        assert _ref().ref_vkr_dump(path.encode(), out.encode()) == 0
        _compare_with_dump(path, json.load(open(out)))


def test_unrepresentable_scenes_are_refused(tmp_path):
    s = scenes.cornell32()
    s.instances[0].transform = s.instances[0].transform.copy()
    s.instances[0].transform[0, 0] = 2.0                      # non-uniform scale
    with pytest.raises(vks.VksError):
        vks.write_vks(str(tmp_path / "x.vks"), s)


def test_reader_errors(tmp_path):
    """the reader's failure cases (vkr.c:781-1101): not a .vks file, unsupported version, truncated, inconsistent offsets"""
    good = open(os.path.join(GOLD, "vks", "alpha_v4.vks"), "rb").read()

    def attempt(data):
        p = str(tmp_path / "t.vks")
        open(p, "wb").write(data)
        return p
    with pytest.raises(vks.VksError):
        vks.read_vks_header(attempt(b"\0" * 64))
    with pytest.raises(vks.VksError):
        vks.read_vks_header(attempt(good[:4] + (9).to_bytes(4, "little") + good[8:]))
    with pytest.raises(vks.VksError):
        vks.read_vks_header(attempt(good[:200]))
    broken = bytearray(good)
    broken[16:24] = (12345).to_bytes(8, "little")             # headerSize
    with pytest.raises(vks.VksError):
        vks.read_vks_header(attempt(bytes(broken)))
    assert vks.read_vkt(str(tmp_path / "missing.vkt")) is None


# ---------------------------------------------------------------- live against the reference's reader
needs_ref = pytest.mark.skipif(not os.path.isfile(REF_LIB), reason="oracle/_ref/libvkr_ref.so not built (needs the reference checkout)")


def _ref():
    lib = C.CDLL(REF_LIB)
    lib.ref_vkr_dump.argtypes = [C.c_char_p, C.c_char_p]
    return lib


@needs_ref
@pytest.mark.parametrize("name", ["cornell32", "textured_test", "two_level_test", "alpha_test"])
@pytest.mark.parametrize("version", [3, 4])
def test_written_files_are_read_identically_by_the_reference(tmp_path, name, version):
    s = getattr(scenes, name)()
    path = str(tmp_path / (name + ".vks"))
    vks.write_vks(path, s, version=version)
    out = str(tmp_path / "dump.json")
    assert _ref().ref_vkr_dump(path.encode(), out.encode()) == 0
    _compare_with_dump(path, json.load(open(out)))


@needs_ref
def test_reference_and_reader_reject_the_same_files(tmp_path):
    good = open(os.path.join(GOLD, "vks", "alpha_v4.vks"), "rb").read()
    shutil.copytree(os.path.join(GOLD, "vks", "alpha_v4_textures"), str(tmp_path / "t_textures"))
    cases = {"magic": b"\1" + good[1:], "version": good[:4] + (7).to_bytes(4, "little") + good[8:], "truncated": good[:300],
             "header_size": good[:16] + (999).to_bytes(8, "little") + good[24:], "good": good}
    for tag, data in cases.items():
        p = str(tmp_path / "t.vks")
        open(p, "wb").write(data)
        rc = _ref().ref_vkr_dump(p.encode(), str(tmp_path / "d.json").encode())
        try:
            vks.read_vks_header(p)
            mine_ok = True
        except vks.VksError:
            mine_ok = False
        assert mine_ok == (rc == 0), tag


def test_vkt_mip_levels_round_trip(tmp_path):
    """.vkt files hold their mip levels back to back behind the level headers (vkr.c:1546-1558): written, read back level by level (RGBA8
    bit for bit, BC1 within the block quantisation, levels below 4 x 4 as one padded block) and carried into the scene's textures"""
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (32, 16, 4)).astype(np.uint8)
    base[..., 3] = 255
    mips, cur = [], base
    while cur.shape[0] > 1 or cur.shape[1] > 1:
        h, w = max(1, cur.shape[0] // 2), max(1, cur.shape[1] // 2)
        cur = cur[:2 * h if cur.shape[0] > 1 else 1, :2 * w if cur.shape[1] > 1 else 1].astype(np.float64)
        cur = np.clip(np.round(cur.reshape(h, cur.shape[0] // h, w, cur.shape[1] // w, 4).mean(axis=(1, 3))), 0, 255).astype(np.uint8)
        mips.append(cur)
    assert [m.shape[:2] for m in mips] == [(16, 8), (8, 4), (4, 2), (2, 1), (1, 1)]
    p = str(tmp_path / "m.vkt")
    vks.write_vkt(p, base, vks.FMT_RGBA8_UNORM, mips=mips)
    l0, fmt, rest = vks.read_vkt(p)
    assert fmt == vks.FMT_RGBA8_UNORM and np.array_equal(l0, base) and len(rest) == 5 and all(np.array_equal(a, b) for a, b in zip(rest, mips))
    flat = np.tile(np.array([[[200, 40, 90, 255]]], np.uint8), (32, 16, 1))
    fm = [np.tile(flat[:1, :1], (max(1, 32 >> l), max(1, 16 >> l), 1)) for l in range(1, 6)]
    vks.write_vkt(p, flat, vks.FMT_BC1_RGB_UNORM, mips=fm)
    l0, fmt, rest = vks.read_vkt(p)
    assert [m.shape for m in rest] == [m.shape for m in fm] and all(np.abs(m.astype(int) - f.astype(int)).max() <= 8 for m, f in zip([l0] + rest, [flat] + fm))
    # through a scene file: the base colour texture of a material keeps its levels
    s = scenes.textured_test()
    tid = next(abi.float_bits(m.base_color[0]) & 0x1FFFFFFF for m in s.materials if abi.float_bits(m.base_color[0]) & 0x80000000)
    t = s.textures[tid]
    tm, cur = [], np.asarray(t.rgba)
    while cur.shape[0] > 1 or cur.shape[1] > 1:
        cur = cur[::2, ::2][:max(1, cur.shape[0] // 2), :max(1, cur.shape[1] // 2)]
        tm.append(np.ascontiguousarray(cur))
    t.mips = tm
    path = str(tmp_path / "t.vks")
    vks.write_vks(path, s)
    r = vks.read_vks(path)
    with_mips = [x for x in r.textures if x.mips]
    assert len(with_mips) >= 1 and len(with_mips[0].levels()) == len(tm) + 1


@needs_ref
def test_mip_mapped_vkt_files_are_read_by_the_reference(tmp_path):
    """pinned by oracle/_ref: .vkt files written with several mip levels are accepted by the reference's own reader, which reports the
    level count, the payload size and the payload offset the writer meant (vkr.c:248-300)"""
    s = scenes.textured_test()
    for t in s.textures:
        lv, cur = [], np.asarray(t.rgba)
        while cur.shape[0] > 1 or cur.shape[1] > 1:
            cur = np.ascontiguousarray(cur[::2, ::2][:max(1, cur.shape[0] // 2), :max(1, cur.shape[1] // 2)])
            lv.append(cur)
        t.mips = lv or None
    path = str(tmp_path / "m.vks")
    vks.write_vks(path, s)
    out = str(tmp_path / "dump.json")
    assert _ref().ref_vkr_dump(path.encode(), out.encode()) == 0
    ref = json.load(open(out))
    _compare_with_dump(path, ref)
    levels = [m["texBaseColor"]["numMipLevels"] for m in ref["materials"] if m["texBaseColor"]]
    assert max(levels) >= 4
    tdir = vks.texture_dir(path)
    for m in ref["materials"]:
        for key, suffix in (("texBaseColor", "_BaseColor.vkt"), ("texNormal", "_Normal.vkt"), ("texSpecular", "_Specular.vkt")):
            if m[key]:
                assert m[key]["dataOffset"] + m[key]["dataSize"] == os.path.getsize(tdir + m["name"] + suffix)
