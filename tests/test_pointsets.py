"""Point sets behind rng_variant: the tables and the oracle's restatement of rendering/pointsets/{sobol,bn_rng,
sample_order}.glsl.  The GLSL cannot be run here (GLM is absent: "parity unpinned" for the lookup code); the TABLES are pinned -- the
Sobol matrices regenerated from the Joe-Kuo direction numbers and the tile inversion derived from them equal the reference's
sobol_tables.h word by word (checked whenever /root/reference is there) and scipy's Sobol' engine point by point."""
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from realtimepathtracingresearchframework_amd import abi, pointsets, scenes

REF_TABLE = "/root/reference/rendering/pointsets/sobol_tables.h"


def lcg_next(s):
    return (s * 1664525 + 1013904223) & 0xFFFFFFFF


def murmur_mix(h, k):
    k = (k * 0xcc9e2d51) & 0xFFFFFFFF
    k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
    k = (k * 0x1b873593) & 0xFFFFFFFF
    h ^= k
    return ((((h << 13) | (h >> 19)) & 0xFFFFFFFF) * 5 + 0xe6546b64) & 0xFFFFFFFF


def murmur_fin(h):
    h ^= h >> 16
    h = (h * 0x85ebca6b) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & 0xFFFFFFFF
    return h ^ (h >> 16)


def sobol_u32(m, index, dim):
    r, j = 0, 0
    while index:
        if index & 1:
            r ^= int(m[dim, j])
        index >>= 1
        j += 1
    return r


# ---------------------------------------------------------------- tables
def test_sobol_matrices_known_answers():
    m = pointsets.sobol_matrices()
    assert m.shape == (1024, 32) and m.dtype == np.uint32
    assert np.array_equal(m[0], (1 << np.arange(31, -1, -1)).astype(np.uint32))  # van der Corput
    # dimension 2 of Joe-Kuo (s = 1, a = 0, m = {1}): v_j = v_{j-1} ^ (v_{j-1} >> 1)
    assert [hex(int(x)) for x in m[1, :4]] == ["0x80000000", "0xc0000000", "0xa0000000", "0xf0000000"]
    # every matrix is upper triangular with a unit diagonal (bit 31-j of word j set, no bit below it): a (0,1)-sequence in base 2
    j = np.arange(32)
    assert np.all((m >> (31 - j)[None, :].astype(np.uint32)) & 1 == 1)
    assert np.all((m & ((np.uint64(1) << (31 - j).astype(np.uint64)) - np.uint64(1)).astype(np.uint32)[None, :]) == 0)


def test_sobol_matrices_against_scipy_engine():
    """scipy walks the same sequence in Gray-code order: its point i is the direct point gray(i)"""
    from scipy.stats import qmc
    m = pointsets.sobol_matrices()
    n, dims = 256, 64
    pts = qmc.Sobol(d=dims, scramble=False, bits=32).random(n)
    gray = np.arange(n) ^ (np.arange(n) >> 1)
    for d in (0, 1, 2, 7, 33, 63):
        mine = pointsets.sobol_points_u32(m, gray, d).astype(np.float64) / 2.0 ** 32
        assert np.array_equal(mine, pts[:, d])


@pytest.mark.skipif(not os.path.exists(REF_TABLE), reason="the reference tree is not on this machine")
def test_sobol_table_equals_the_references_table():
    txt = open(REF_TABLE).read()
    i0, i1 = txt.index("SobolMatrix"), txt.index("SobolInversion_1_0")
    ref_m = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})U", txt[i0:i1])], dtype=np.uint32)
    body = txt[txt.index("{", i1) + 1:txt.index("}", i1)]
    ref_inv = np.array([int(x) for x in re.findall(r"\d+", body)], dtype=np.uint32)
    t = pointsets.sobol_table()
    assert ref_m.size == 1024 * 32 and ref_inv.size == 256 * 256
    assert np.array_equal(t[:1024 * 32], ref_m)
    assert np.array_equal(t[1024 * 32:], ref_inv)


REF_TOOL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "oracle", "_ref", "prepare_sobol")


def _run_ref_tool(bits, tile, dx, dy):
    import subprocess
    txt = subprocess.run([REF_TOOL, str(bits), str(tile), str(dx), str(dy)], capture_output=True, text=True, check=True).stdout
    i1 = txt.index("SobolInversion_")
    m = np.array([int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]{8})U", txt[:i1])], dtype=np.uint32)
    body = txt[txt.index("{", i1) + 1:txt.index("}", i1)]
    return m, np.array([int(x) for x in re.findall(r"\d+", body)], dtype=np.uint32), txt


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="oracle/_ref/prepare_sobol is built from the reference tree (oracle/Makefile)")
def test_sobol_table_equals_the_output_of_the_references_generator():
    """pinned by oracle/_ref: rendering/tools/prepare_sobol.cpp compiled unmodified prints the tables the shaders read; the package's
    regenerated matrices and the derived tile inversion equal them word by word — for the shipped (32 bits, 256 tile, dims 0 / 1) table
    and for other tile sizes and dimension pairs of the same tool"""
    m, inv, txt = _run_ref_tool(32, 256, 0, 1)
    t = pointsets.sobol_table()
    assert m.size == 1024 * 32 and inv.size == 256 * 256 and "Zeros: 1" in txt
    assert np.array_equal(t[:1024 * 32], m) and np.array_equal(t[1024 * 32:], inv)
    mine = pointsets.sobol_matrices()
    for tile, dx, dy in ((64, 0, 1), (16, 1, 0), (128, 0, 1)):
        _, inv, txt = _run_ref_tool(32, tile, dx, dy)
        assert "Zeros: 1" in txt and np.array_equal(pointsets.sobol_tile_inversion(mine, tile=tile, dim_x=dx, dim_y=dy), inv)


def test_tile_inversion_is_the_inverse_of_the_first_two_dimensions():
    m = pointsets.sobol_matrices()
    inv = pointsets.sobol_tile_inversion(m)
    assert sorted(inv.tolist()) == list(range(65536))  # the first 2^16 points hit every cell of the 256 x 256 grid once
    i = np.arange(65536)
    x = pointsets.sobol_points_u32(m, i, 0) >> 24
    y = pointsets.sobol_points_u32(m, i, 1) >> 24
    assert np.array_equal(inv[y.astype(np.int64) * 256 + x.astype(np.int64)], i.astype(np.uint32))


def test_bn_stand_in_table_layout():
    t = pointsets.white_noise_bn_table(3)
    assert t.nbytes == abi.BN_TABLE_MIN_BYTES and t.max() < 256
    seq = t[:65536].reshape(256, 256)
    for d in (0, 5, 200):  # every dimension: each 8-bit value once
        assert sorted(seq[:, d].tolist()) == list(range(256))


# ---------------------------------------------------------------- the oracle's lookups
@pytest.fixture(scope="module")
def osc():
    s = O.OracleScene(scenes.cornell32())
    yield s
    s.close()


def test_sobol_draws_are_scrambled_sequence_points(osc):
    m = pointsets.sobol_matrices()
    osc.set_rng_variant(abi.RNG_VARIANT_SOBOL, pointsets.sobol_table())
    W = 40
    for (si, fo, px, py) in [(0, 0, 0, 0), (5, 9, 3, 7), (1023, 77, 39, 11)]:
        dims = [0, 1, 3, 2, 7, -1]
        vals, index = osc.pointset_probe(si, fo, 123, px, py, W, 14, dims)
        assert index == si
        s = murmur_fin(murmur_mix(murmur_mix(0, px + py * W), fo))  # get_lcg_rng(frame_offset, 0, pixel): per-pixel scramble
        for k, d in enumerate(dims):
            s = lcg_next(s)
            u = sobol_u32(m, si, (14 + d) & 1023) ^ s
            assert vals[k] == np.float32(np.ldexp(np.float32(u), -32))
    # dimensions wrap at 1024
    a, _ = osc.pointset_probe(9, 1, 0, 2, 2, W, 0, [5])
    b, _ = osc.pointset_probe(9, 1, 0, 2, 2, W, 1024, [5])
    assert a[0] == b[0]
    osc.set_rng_variant(abi.RNG_VARIANT_UNIFORM)


def test_z_sobol_assigns_every_pixel_of_a_tile_its_own_point(osc):
    """Z_ORDER_SHUFFLING: within a 256 x 256 tile, sample k of the pixels uses the points [k 2^16, (k+1) 2^16) once each, and the
    point of a pixel is the one whose first two coordinates land on the pixel's (shuffled) cell"""
    m = pointsets.sobol_matrices()
    osc.set_rng_variant(abi.RNG_VARIANT_Z_SBL, pointsets.sobol_table())
    W = 512
    for k in (0, 3):
        seen = set()
        for (px, py) in sorted({(x, y) for x in range(256, 512, 17) for y in range(0, 256, 13)} | {(256, 0), (511, 255), (300, 200)}):
            _, index = osc.pointset_probe(k, 4, 0, px, py, W, 0, [0])
            assert k * 65536 <= index < (k + 1) * 65536
            assert index not in seen
            seen.add(index)
    # sample 0: the index is the pixel's position on the shuffled Z curve, so the pixels of an aligned 2 x 2 (4 x 4) block share an
    # aligned run of 4 (16) consecutive points -- which the (0, m, 2)-net property spreads over the 4 quadrants (16 cells)
    for size in (2, 4):
        idx = [osc.pointset_probe(0, 0, 0, 8 + dx, 20 + dy, W, 0, [0])[1] for dy in range(size) for dx in range(size)]
        n = size * size
        assert len(set(idx)) == n and len({i // n for i in idx}) == 1
        sh = 32 - (size.bit_length() - 1)
        assert len({(sobol_u32(m, i, 0) >> sh, sobol_u32(m, i, 1) >> sh) for i in idx}) == n
    # the scramble is per tile: two pixels of one tile share it, the leading-bit fix-up applies to dimensions 0 and 1 only
    s = lcg_next(murmur_fin(murmur_mix(murmur_mix(0, (300 >> 8) + (200 >> 8) * (W >> 8)), 4)))
    v, index = osc.pointset_probe(2, 4, 0, 300, 200, W, 0, [2])
    assert v[0] == np.float32(np.ldexp(np.float32(sobol_u32(m, index, 2) ^ s), -32))
    v, index = osc.pointset_probe(2, 4, 0, 300, 200, W, 0, [1])
    u = sobol_u32(m, index, 1) ^ s
    u ^= (u << 8) & 0xFFFFFFFF
    assert v[0] == np.float32(np.ldexp(np.float32(u), -32))
    osc.set_rng_variant(abi.RNG_VARIANT_UNIFORM)


def test_blue_noise_draws(osc):
    t = pointsets.white_noise_bn_table(2)
    osc.set_rng_variant(abi.RNG_VARIANT_BN, t)
    seq, keys = t[:65536].reshape(256, 256), t[65536:].reshape(128 * 128, 8)
    # sampleID = frame_id + 13 frame_offset = 0: no mirroring, no shift; dimension d < 8: value = seq[0, d] ^ keys[pixel, d]
    v, index = osc.pointset_probe(7, 0, 0, 5 + 128, 9, 640, 0, list(range(8)))
    assert index == 0
    pix = 5 + 9 * 128
    assert np.array_equal(v, ((seq[0, :8] ^ keys[pix, :8]).astype(np.float32) + 0.5) / 256.0)
    # dimension 8..15: the mask one pixel to the right, same sequence dimensions (BN_OPTIMIZED_DIMENSION_REPEAT)
    v, _ = osc.pointset_probe(7, 0, 0, 5, 9, 640, 8, list(range(8)))
    assert np.array_equal(v, ((seq[0, :8] ^ keys[pix + 1, :8]).astype(np.float32) + 0.5) / 256.0)
    # the sample index passed to GET_RNG is ignored (bn_rng.glsl:112): all samples of one frame repeat the draws; the next frame differs
    a, _ = osc.pointset_probe(0, 0, 1, 5, 9, 640, 6, [0, 1, 2])
    b, _ = osc.pointset_probe(3, 0, 1, 5, 9, 640, 6, [0, 1, 2])
    c, i2 = osc.pointset_probe(0, 0, 2, 5, 9, 640, 6, [0, 1, 2])
    assert np.array_equal(a, b) and not np.array_equal(a, c) and i2 == 2
    assert np.all((v > 0) & (v < 1))
    osc.set_rng_variant(abi.RNG_VARIANT_UNIFORM)


def test_rng_variant_rejects_short_tables(osc):
    assert O.lib().orc_scene_set_rng_variant(osc.h, abi.RNG_VARIANT_SOBOL, None, 0) != 0
    assert O.lib().orc_scene_set_rng_variant(osc.h, 7, None, 0) != 0


def test_low_discrepancy_points_beat_the_uniform_generator_on_a_smooth_image():
    """what the point sets are for: the same scene, the same sample count, less error (against a 512 spp image of the uniform generator)"""
    sc = scenes.cornell32()
    osc = O.OracleScene(sc)
    W, H = 32, 24
    ref, _ = osc.render(W, H, 512, variant=abi.VARIANT_SIMPLE)
    ref = ref[..., :3]
    err = {}
    for name, var, table in (("uniform", abi.RNG_VARIANT_UNIFORM, None), ("sobol", abi.RNG_VARIANT_SOBOL, pointsets.sobol_table()),
                             ("z_sobol", abi.RNG_VARIANT_Z_SBL, pointsets.sobol_table())):
        osc.set_rng_variant(var, table)
        img, _ = osc.render(W, H, 16, variant=abi.VARIANT_SIMPLE, frame_offset=3)
        assert np.isfinite(img).all()
        err[name] = float(np.sqrt(np.mean((img[..., :3] - ref) ** 2)))
    osc.close()
    assert err["sobol"] < err["uniform"] and err["z_sobol"] < err["uniform"], err


# ------------------------------------------------------------------ the optimised blue-noise tables (reference data, read where they lie)
REF_BN = "/root/reference/rendering/pointsets/bn_tables.h"
needs_ref_bn = pytest.mark.skipif(not os.path.exists(REF_BN), reason="the reference's bn_tables.h is not on this machine")


def _low_frequency_share(img, radius=10):
    """share of the spectrum's energy (mean removed) below `radius` cycles per tile: ~ pi r^2 / n^2 for white noise"""
    n = img.shape[0]
    f = np.fft.fftshift(np.abs(np.fft.fft2(img - img.mean())) ** 2)
    yy, xx = np.mgrid[-n // 2:n // 2, -n // 2:n // 2]
    return float(f[np.sqrt(xx * xx + yy * yy) < radius].sum() / f.sum())


@needs_ref_bn
def test_the_references_blue_noise_header_parses_into_bndata():
    """bn_data.h:12-27 / vulkan/pointsets/render_bn.cpp:84-104: sobol_spp_d, then scrambling (+ ranking above 1 spp) keys per optimised spp"""
    arrs = pointsets.read_bn_tables_header(REF_BN)
    assert set(pointsets.BN_HEADER_ARRAYS) <= set(arrs) and "rankingTile_yx_d_1spp" in arrs
    t = pointsets.bn_table_from_header(REF_BN)
    assert t.dtype == np.uint32 and t.size == 256 * 256 + 7 * 128 * 128 * 8 and t.nbytes >= abi.BN_TABLE_MIN_BYTES and int(t.max()) == 255
    at = 0
    for name in pointsets.BN_HEADER_ARRAYS:
        assert np.array_equal(t[at:at + arrs[name].size], arrs[name]), name
        at += arrs[name].size
    assert not arrs["rankingTile_yx_d_1spp"].any() and int(arrs["rankingTile_yx_d_4spp"].max()) == 3 and int(arrs["rankingTile_yx_d_16spp"].max()) == 15
    # every dimension of the 256-sample sequence is a permutation of the 256 byte values (an Owen-scrambled (0, 8, 1)-net in base 2)
    seq = t[:65536].reshape(256, 256)
    assert all(np.array_equal(np.sort(seq[:, d]), np.arange(256)) for d in range(256))


@needs_ref_bn
def test_cpp_reader_of_the_blue_noise_header_equals_the_python_reader(tmp_path):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "bn.cpp"
    src.write_text('#include "realtimepathtracingresearchframework_amd/host/pointsets.hpp"\n'
                   'int main(int argc, char **argv) { try { const std::vector<uint32_t> t = rptr::read_bn_table(argv[1]);\n'
                   '  std::FILE *f = std::fopen(argv[2], "wb"); std::fwrite(t.data(), 4, t.size(), f); std::fclose(f); return 0; }\n'
                   '  catch (const std::exception &e) { std::fprintf(stderr, "%s\\n", e.what()); return 1; } }\n')
    exe = str(tmp_path / "bn")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + root, str(src), "-o", exe])
    out = str(tmp_path / "BNData.u32")
    subprocess.check_call([exe, REF_BN, out])
    t = pointsets.bn_table_from_header(REF_BN)
    assert np.array_equal(np.fromfile(out, dtype="<u4"), t)
    out2 = str(tmp_path / "again.u32")   # ... and the raw words are accepted as they are
    subprocess.check_call([exe, out, out2])
    assert np.array_equal(np.fromfile(out2, dtype="<u4"), t)
    bad = tmp_path / "short.h"
    bad.write_text("static const int sobol_256spp_256d[4] = {1,2,3,4};\n")
    p = subprocess.run([exe, str(bad), out2], capture_output=True, text=True)
    assert p.returncode == 1 and "expected 65536" in p.stderr


@needs_ref_bn
def test_oracle_draws_from_the_real_tables_are_blue_noise_in_screen_space(osc):
    """The lookup (oracle/oshade.h = bn_rng.glsl:31-75) fed with the REAL tables: the 1 spp draws of a dimension over a 128 x 128 tile are a
    blue-noise mask -- almost no energy at low frequencies (5e-5 of the spectrum below 10 cycles per tile; white noise: 2e-2). Any mistake
    in the layout (pixel = x + 128 y, 8 keys per pixel, BNData's member order) or in the mask shifts of later dimensions / frames turns
    the mask white. The stand-in table, read by the same code, is white by construction."""
    real, stand_in = pointsets.bn_table_from_header(REF_BN), pointsets.white_noise_bn_table(2)
    W = 640

    def mask(frame_id, set_dim, dim, x0=0, y0=0):
        img = np.zeros((128, 128), np.float32)
        for y in range(128):
            for x in range(128):
                img[y, x] = osc.pointset_probe(0, 0, frame_id, x0 + x, y0 + y, W, set_dim, [dim])[0][0]
        return img
    osc.set_rng_variant(abi.RNG_VARIANT_BN, real)
    shares = {}
    for (frame_id, set_dim, dim, x0, y0) in ((0, 0, 0, 0, 0), (0, 0, 7, 128, 0), (0, 8, 3, 0, 128), (1, 0, 2, 256, 128), (3, 6, 5, 0, 0)):
        m = mask(frame_id, set_dim, dim, x0, y0)
        assert m.min() > 0.0 and m.max() < 1.0
        assert abs(float(m.mean()) - 0.5) < 0.01 and abs(float(m.std()) - 12 ** -0.5) < 0.01   # uniform over the tile
        shares[(frame_id, set_dim, dim)] = _low_frequency_share(m)
    osc.set_rng_variant(abi.RNG_VARIANT_BN, stand_in)
    white = _low_frequency_share(mask(0, 0, 0))
    osc.set_rng_variant(abi.RNG_VARIANT_UNIFORM)
    print("low-frequency share of the draws' spectrum: real tables", {k: round(v, 6) for k, v in shares.items()}, " stand-in %.4f" % white)
    assert max(shares.values()) < 1e-3 and white > 5e-3, (shares, white)
