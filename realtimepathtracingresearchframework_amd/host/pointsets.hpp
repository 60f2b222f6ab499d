// pointsets.hpp -- host side of rng_variant (librender/render_params.glsl.h:34-37): the tables rptr_hip_set_rng_variant takes, laid out as
// the reference uploads them (vulkan/pointsets/render_sobol.cpp:84-104, render_bn.cpp:78-126). The C++ twin of pointsets.py.
//
//   SobolData: the Joe-Kuo generator matrices from package data (data/sobol_matrices_1024x32.u32, tools/gen_sobol_matrices.py) + the
//              256 x 256 tile inversion derived from the first two of them (what rendering/tools/prepare_sobol.cpp:36-58 prints).
//   BNData:    the reference's blue-noise tables are the published output of an optimiser and are handed over as a file
//              (--bn-table: the reference's bn_tables.h itself, parsed at run time, or raw little-endian uint32 words of BNData); without one, a table of the same layout without the optimisation
//              (digitally shifted Sobol' values, hashed scrambling keys): white-noise instead of blue-noise error distribution.
#pragma once
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rptr_hip.h"

namespace rptr {

inline std::vector<uint32_t> read_u32_file(const std::string &path) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> w((size_t)n / 4);
    const size_t got = std::fread(w.data(), 4, w.size(), f);
    std::fclose(f);
    if (got != w.size()) throw std::runtime_error("short read: " + path);
    return w;
}

inline uint32_t sobol_point_u32(const uint32_t *matrices, uint32_t index, uint32_t dim) {
    uint32_t r = 0;
    for (uint32_t j = 0; index; index >>= 1, ++j)
        if (index & 1u) r ^= matrices[dim * 32u + j];
    return r;
}

// SobolData words: matrix[1024 * 32] then tile_invert_1_0[256 * 256]
inline std::vector<uint32_t> sobol_table(const std::string &matrices_path) {
    std::vector<uint32_t> t = read_u32_file(matrices_path);
    if (t.size() != 1024u * 32u) throw std::runtime_error(matrices_path + ": expected 1024 x 32 words");
    t.resize(RPTR_SOBOL_TABLE_BYTES / 4, 0u);
    uint32_t *inv = t.data() + 1024u * 32u;
    for (uint32_t i = 0; i < 256u * 256u; ++i) { // cell of the 256 x 256 grid the i-th point falls into -> i
        const uint32_t x = sobol_point_u32(t.data(), i, 0) >> 24, y = sobol_point_u32(t.data(), i, 1) >> 24;
        inv[y * 256u + x] = i;
    }
    return t;
}

inline uint32_t mix32(uint32_t x) { // (a finaliser-style hash; only the stand-in table uses it)
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

// BNData prefix without the blue-noise optimisation: sobol_spp_d[256 * 256] + tile_scrambling_yx_d_1spp[128 * 128 * 8]
inline std::vector<uint32_t> white_noise_bn_table(const std::string &matrices_path, uint32_t seed = 1) {
    const std::vector<uint32_t> m = read_u32_file(matrices_path);
    if (m.size() != 1024u * 32u) throw std::runtime_error(matrices_path + ": expected 1024 x 32 words");
    std::vector<uint32_t> t(RPTR_BN_TABLE_MIN_BYTES / 4);
    for (uint32_t d = 0; d < 256u; ++d) {
        const uint32_t shift = mix32(seed * 0x9e3779b9u + d) & 255u;
        for (uint32_t i = 0; i < 256u; ++i) t[i * 256u + d] = (sobol_point_u32(m.data(), i, d) >> 24) ^ shift;
    }
    for (uint32_t k = 0; k < 128u * 128u * 8u; ++k) t[256u * 256u + k] = mix32(seed * 0x85ebca6bu + 0x10000u + k) & 255u;
    return t;
}

// BNData (bn_data.h:12-27) from the C arrays of a header of blue-noise tables (the reference's rendering/pointsets/bn_tables.h), read at
// run time: sobol_256spp_256d, then the scrambling (and, above 1 spp, ranking) keys per optimised sample count. = pointsets.py
// bn_table_from_header, word for word.
inline std::vector<uint32_t> bn_table_from_header(const std::string &path) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string txt;
    char buf[1 << 16];
    for (size_t n; (n = std::fread(buf, 1, sizeof(buf), f)) > 0;) txt.append(buf, n);
    std::fclose(f);
    auto array_of = [&](const std::string &name, size_t want, bool required, std::vector<uint32_t> &out) -> bool {
        const std::string key = " " + name + "[";
        const size_t at = txt.find(key);
        if (at == std::string::npos) {
            if (required) throw std::runtime_error(path + ": no array `" + name + "`");
            return false;
        }
        const size_t open = txt.find('{', at), close = txt.find('}', open == std::string::npos ? at : open);
        if (open == std::string::npos || close == std::string::npos) throw std::runtime_error(path + ": `" + name + "` has no initialiser");
        out.clear();
        out.reserve(want);
        uint32_t v = 0;
        bool in_number = false;
        for (size_t i = open + 1; i < close; ++i) {
            const char ch = txt[i];
            if (ch >= '0' && ch <= '9') {
                v = v * 10u + uint32_t(ch - '0');
                in_number = true;
            } else if (in_number) {
                out.push_back(v);
                v = 0;
                in_number = false;
            }
        }
        if (in_number) out.push_back(v);
        if (out.size() != want) throw std::runtime_error(path + ": `" + name + "` has " + std::to_string(out.size()) + " values, expected " + std::to_string(want));
        for (uint32_t w : out)
            if (w > 255u) throw std::runtime_error(path + ": `" + name + "` holds values above 255");
        return true;
    };
    static const char *names[8] = {"sobol_256spp_256d",        "scramblingTile_yx_d_1spp",  "scramblingTile_yx_d_4spp",   "rankingTile_yx_d_4spp",
                                   "scramblingTile_yx_d_16spp", "rankingTile_yx_d_16spp",    "scramblingTile_yx_d_256spp", "rankingTile_yx_d_256spp"};
    std::vector<uint32_t> t, part;
    for (int k = 0; k < 8; ++k) {
        array_of(names[k], k == 0 ? 256u * 256u : 128u * 128u * 8u, true, part);
        t.insert(t.end(), part.begin(), part.end());
    }
    if (array_of("rankingTile_yx_d_1spp", 128u * 128u * 8u, false, part)) // bn_data.h:17: no member, because the keys are { 0 }
        for (uint32_t w : part)
            if (w != 0u) throw std::runtime_error(path + ": the 1 spp ranking keys are not all zero: not the table set BNData expects");
    return t;
}

// --bn-table <file>: a C header of tables (above) or the raw little-endian words of BNData
inline std::vector<uint32_t> read_bn_table(const std::string &path) {
    const size_t dot = path.rfind('.');
    const std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".h" || ext == ".hpp" || ext == ".cpp" || ext == ".c" || ext == ".inl") return bn_table_from_header(path);
    return read_u32_file(path);
}

inline int rng_variant_from_name(const std::string &s) { // RNG_VARIANT_NAMES (render_params.glsl.h:39-43), any case, -1: unknown
    std::string u;
    for (char ch : s) u += (char)std::toupper((unsigned char)ch);
    if (u.find("Z_SBL") != std::string::npos || u.find("Z-SOBOL") != std::string::npos || u.find("Z_SOBOL") != std::string::npos) return RPTR_RNG_VARIANT_Z_SBL;
    if (u.find("SOBOL") != std::string::npos) return RPTR_RNG_VARIANT_SOBOL;
    if (u.find("BN") != std::string::npos || u.find("BLUE") != std::string::npos) return RPTR_RNG_VARIANT_BN;
    if (u.find("UNIFORM") != std::string::npos || u.find("LCG") != std::string::npos) return RPTR_RNG_VARIANT_UNIFORM;
    return -1;
}

} // namespace rptr
