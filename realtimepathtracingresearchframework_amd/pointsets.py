"""Tables of the point sets behind `rng_variant` (librender/render_params.glsl.h:34-37), laid out as the reference uploads them.

* Sobol / Z-Sobol: `SobolData` (rendering/pointsets/sobol_data.h:13-17) = the 1024 x 32 generator matrices of the Joe-Kuo Sobol'
  sequence (package data, tools/gen_sobol_matrices.py) followed by the 256 x 256 tile inversion table, which is derived from the first
  two matrices here exactly as rendering/tools/prepare_sobol.cpp:36-58 derives it.
* blue noise: `BNData` (bn_data.h:12-27). The reference's tables are the published output of an optimiser (Heitz et al. 2019:
  an Owen-scrambled 256-sample sequence, and per-pixel scrambling keys optimised for a blue-noise error distribution); they are data
  an integration hands to `rptr_hip_set_rng_variant` as they are. `white_noise_bn_table` builds a table of the same layout without the
  optimisation (a digitally shifted Sobol' sequence and random keys): every pixel still gets a well-distributed sequence, the error is
  white instead of blue in screen space. Tests use it for parity of the lookup arithmetic. The optimised tables themselves stay where
  they are: `bn_table_from_header` reads the C arrays of the reference's `rendering/pointsets/bn_tables.h` at run time (`RPTR_BN_DATA=<that
  file>`, CLI `--bn-table <that file>`), the way `sky_fit.py` reads the Hosek-Wilkie coefficient headers.

Host-side only: nothing here touches the GPU.
"""
import os

import numpy as np

from . import abi

SOBOL_DIMS, SOBOL_BITS, SOBOL_TILE = 1024, 32, 256
BN_SAMPLES, BN_DIMS, BN_SCR_DIMS, BN_TILE = 256, 256, 8, 128

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "sobol_matrices_1024x32.u32")


def sobol_matrices():
    """(1024, 32) uint32: word j of dimension d is XORed into the point when bit j of the index is set (sobol.glsl:74-83)"""
    m = np.fromfile(_DATA, dtype="<u4")
    if m.size != SOBOL_DIMS * SOBOL_BITS:
        raise RuntimeError("%s: expected %d words, found %d" % (_DATA, SOBOL_DIMS * SOBOL_BITS, m.size))
    return m.reshape(SOBOL_DIMS, SOBOL_BITS).astype(np.uint32)


def sobol_points_u32(matrices, indices, dim):
    """coordinate `dim` of the points `indices` as 32-bit fixed point"""
    idx = np.asarray(indices, dtype=np.uint64)
    out = np.zeros(idx.shape, dtype=np.uint32)
    for j in range(SOBOL_BITS):
        out ^= np.where((idx >> np.uint64(j)) & np.uint64(1), matrices[dim, j], np.uint32(0)).astype(np.uint32)
    return out


def sobol_tile_inversion(matrices, tile=SOBOL_TILE, dim_x=0, dim_y=1):
    """tile_invert_1_0[y * tile + x] = the index i < tile^2 whose first two coordinates fall into cell (x, y) of the tile x tile grid
    (rendering/tools/prepare_sobol.cpp:36-58; a bijection because the two dimensions form a (0, 2 log2 tile, 2)-net)"""
    bits = int(tile - 1).bit_length()
    i = np.arange(tile * tile, dtype=np.uint64)
    x = sobol_points_u32(matrices, i, dim_x) >> np.uint32(32 - bits)
    y = sobol_points_u32(matrices, i, dim_y) >> np.uint32(32 - bits)
    table = np.zeros(tile * tile, dtype=np.uint32)
    table[(y.astype(np.int64) * tile + x.astype(np.int64))] = i.astype(np.uint32)
    return table


_sobol_cache = None


def sobol_table():
    """SobolData as uint32 words: matrix[1024 * 32] + tile_invert_1_0[256 * 256] (RPTR_SOBOL_TABLE_BYTES)"""
    global _sobol_cache
    if _sobol_cache is None:
        m = sobol_matrices()
        _sobol_cache = np.ascontiguousarray(np.concatenate([m.reshape(-1), sobol_tile_inversion(m)]), dtype=np.uint32)
        assert _sobol_cache.nbytes == abi.SOBOL_TABLE_BYTES
    return _sobol_cache


def white_noise_bn_table(seed=1):
    """BNData prefix (sobol_spp_d[256 * 256] + tile_scrambling_yx_d_1spp[128 * 128 * 8]) without the blue-noise optimisation"""
    rng = np.random.RandomState(seed)
    m = sobol_matrices()
    seq = np.zeros((BN_SAMPLES, BN_DIMS), dtype=np.uint32)
    i = np.arange(BN_SAMPLES, dtype=np.uint64)
    for d in range(BN_DIMS):
        seq[:, d] = (sobol_points_u32(m, i, d) >> np.uint32(24)) ^ np.uint32(rng.randint(0, 256))
    keys = rng.randint(0, 256, size=BN_TILE * BN_TILE * BN_SCR_DIMS).astype(np.uint32)
    t = np.ascontiguousarray(np.concatenate([seq.reshape(-1), keys]), dtype=np.uint32)
    assert t.nbytes == abi.BN_TABLE_MIN_BYTES
    return t


# the arrays of bn_tables.h in the order BNData lays them out (bn_data.h:12-27; the 1 spp ranking keys are all zero and have no member)
BN_HEADER_ARRAYS = ("sobol_256spp_256d", "scramblingTile_yx_d_1spp", "scramblingTile_yx_d_4spp", "rankingTile_yx_d_4spp",
                    "scramblingTile_yx_d_16spp", "rankingTile_yx_d_16spp", "scramblingTile_yx_d_256spp", "rankingTile_yx_d_256spp")


def read_bn_tables_header(path):
    """{array name: uint32 words} of every `static const int name[...] = {...};` in a C header of blue-noise tables
    (the reference's rendering/pointsets/bn_tables.h, or Heitz et al.'s published samplerBlueNoiseErrorDistribution_*.cpp)"""
    import re
    with open(path, "r") as f:
        txt = f.read()
    out = {}
    for m in re.finditer(r"static\s+const\s+int\s+(\w+)\s*\[[^\]]*\]\s*=\s*\{([^}]*)\}", txt):
        out[m.group(1)] = np.array(re.findall(r"\d+", m.group(2)), dtype=np.uint32)
    return out


def bn_table_from_header(path):
    """BNData (bn_data.h:12-27) as uint32 words from the arrays of `path`: what render_vulkan.cpp uploads for RNG_VARIANT_BN"""
    arrs = read_bn_tables_header(path)
    sizes = {"sobol_256spp_256d": BN_SAMPLES * BN_DIMS}
    parts = []
    for name in BN_HEADER_ARRAYS:
        if name not in arrs:
            raise RuntimeError("%s: no array `%s`" % (path, name))
        want = sizes.get(name, BN_TILE * BN_TILE * BN_SCR_DIMS)
        if arrs[name].size != want:
            raise RuntimeError("%s: `%s` has %d values, expected %d" % (path, name, arrs[name].size, want))
        if int(arrs[name].max()) > 255:
            raise RuntimeError("%s: `%s` holds values above 255" % (path, name))
        parts.append(arrs[name])
    rank1 = arrs.get("rankingTile_yx_d_1spp")
    if rank1 is not None and np.any(rank1 != 0):   # bn_data.h:17: BNData has no such member because the keys are { 0 }
        raise RuntimeError("%s: the 1 spp ranking keys are not all zero: not the table set BNData expects" % path)
    t = np.ascontiguousarray(np.concatenate(parts), dtype=np.uint32)
    assert t.nbytes >= abi.BN_TABLE_MIN_BYTES
    return t


def default_table(rng_variant, seed=1):
    """the table `HipBackend.set_rng_variant` uploads when the caller passes none"""
    if rng_variant in (abi.RNG_VARIANT_SOBOL, abi.RNG_VARIANT_Z_SBL):
        return sobol_table()
    if rng_variant == abi.RNG_VARIANT_BN:
        if os.environ.get("RPTR_BN_DATA"):   # the optimised tables, read from the header an installation of the reference holds
            return bn_table_from_header(os.environ["RPTR_BN_DATA"])
        return white_noise_bn_table(seed)
    return None
