// TEST INFRASTRUCTURE -- CPU oracle: restatement of the reference's shader
// library for the PT_MEGAKERNEL path, one function per reference function, in
// the reference's evaluation order (GLSL: call arguments left to right).
// All file:line citations are relative to /root/reference.
//
// Parity status: the reference's own tests hold no golden values for these
// functions. Pinned here: the RNG known-answer recorded in
// SURVEY 8(a2); the sky/sun host fit via oracle/_ref (compiled from the
// reference's sky_model.cpp). Everything else on this page is "parity
// unpinned" -- checked by domain properties in tests/ only.
#pragma once
#include "ovec.h"
#include <vector>
#include "../include/rptr_hip.h"

namespace orc {

#ifndef M_PIf
#define M_PIf 3.14159265358979323846f
#endif
#ifndef M_1_PIf
#define M_1_PIf 0.318309886183790671538f
#endif
#define ORC_EPSILON 0.0001f // rendering/defaults.glsl:14-16

// ---------------------------------------------------------------- RNG
// rendering/pointsets/hashing.glsl:11-28
static inline uint32_t murmur_hash3_mix(uint32_t hash, uint32_t k) {
    const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u, r1 = 15, r2 = 13, m = 5, n = 0xe6546b64u;
    k *= c1;
    k = (k << r1) | (k >> (32 - r1));
    k *= c2;
    hash ^= k;
    hash = ((hash << r2) | (hash >> (32 - r2))) * m + n;
    return hash;
}
// rendering/pointsets/hashing.glsl:30-39
static inline uint32_t murmur_hash3_finalize(uint32_t hash) {
    hash ^= hash >> 16;
    hash *= 0x85ebca6bu;
    hash ^= hash >> 13;
    hash *= 0xc2b2ae35u;
    hash ^= hash >> 16;
    return hash;
}
struct LCGRand {
    uint32_t state;
};
// rendering/pointsets/lcg_rng.glsl:15-21
static inline uint32_t lcg_random(LCGRand &rng) {
    rng.state = rng.state * 1664525u + 1013904223u;
    return rng.state;
}
// rendering/pointsets/lcg_rng.glsl:23-26  (may return exactly 1.0f: reproduced)
static inline float lcg_randomf(LCGRand &rng) { return ldexpf((float)lcg_random(rng), -32); }
// rendering/pointsets/lcg_rng.glsl:28-39
static inline LCGRand get_lcg_rng(uint32_t index, uint32_t frame, uint32_t linear) {
    LCGRand rng;
    rng.state = murmur_hash3_mix(frame, linear);
    rng.state = murmur_hash3_mix(rng.state, index);
    rng.state = murmur_hash3_finalize(rng.state);
    return rng;
}
static inline LCGRand get_lcg_rng(uint32_t index, uint32_t frame, uint32_t px, uint32_t py, uint32_t dimx) {
    return get_lcg_rng(index, frame, px + py * dimx);
}
// rendering/defaults.glsl:29-35 (ordered: x first, then y)
static inline vec2 random_float2(LCGRand &rng) {
    vec2 r;
    r.x = lcg_randomf(rng);
    r.y = lcg_randomf(rng);
    return r;
}

// ------------------------------------------------------------------ point sets (RBO rng_variant, librender/render_params.glsl.h:34-37)
// RANDOM_STATE / RANDOM_FLOAT1 / RANDOM_SET_DIM / RANDOM_SHIFT_DIM / GET_RNG of rendering/pointsets/{lcg_rng,sobol,bn_rng}.glsl
// behind one type (rendering/pointsets/selected_rng.glsl picks one at shader compile time; here `variant` does at run time).
struct PointSetTable { // the buffer at RANDOM_NUMBERS_BIND_POINT: SobolData (sobol_data.h:13-17) or BNData (bn_data.h:12-27)
    int variant = 0;   // RPTR_RNG_VARIANT_*
    std::vector<uint32_t> words;
};
enum { SobolData_Dimensions = 1024, SobolData_MatrixSize = 32, SobolData_TileSize = 256 };
enum { BNData_SampleCount = 256, BNData_Dimensions = 256, BNData_ScramblingDimensions = 8, BNData_TileSize = 128 };
struct RandomState {
    const PointSetTable *table = nullptr; // nullptr: uniform
    LCGRand lcg;          // uniform: the generator; Sobol: the scramble
    uint32_t index = 0;   // Sobol: point index; blue noise: sampleID
    uint32_t pixel = 0;   // blue noise: pixelID in the tile
    int32_t dimension = 0;
    int variant() const { return table ? table->variant : RPTR_RNG_VARIANT_UNIFORM; }
};
static inline uint32_t part1by1(uint32_t x) { // rendering/util.glsl:156-163
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
static inline int find_msb(uint32_t x) { return x ? 31 - __builtin_clz(x) : -1; }
// rendering/pointsets/sample_order.glsl:22-73, all parameters kept
static inline uint32_t morton_sample_id(uint32_t sample_id, uint32_t pixel_x, uint32_t pixel_y, uint32_t tile_x, uint32_t tile_y, bool hash_tile_id,
                                        bool hash_sample_id) {
    uint32_t padded_x = 1u << find_msb(tile_x), padded_y = 1u << find_msb(tile_y);
    if (padded_x != tile_x) padded_x <<= 1;
    if (padded_y != tile_y) padded_y <<= 1;
    const uint32_t pcount = padded_x * padded_y;
    const uint32_t ex = part1by1(pixel_x), ey = part1by1(pixel_y);
    uint32_t linear = (ey << 1) + ex;
    const uint32_t min_dim_mask = (padded_x - 1u) & (padded_y - 1u);
    const uint32_t interleaved_mask = (min_dim_mask + 1u) * (min_dim_mask + 1u) - 1u;
    linear &= interleaved_mask;
    linear |= ((pixel_x | pixel_y) & ~min_dim_mask) * (min_dim_mask + 1u);
    if (!hash_tile_id) linear &= pcount - 1u;
    uint32_t scrambled = linear;
    uint32_t swap_vec = ex ^ ey;
    swap_vec |= swap_vec << 1;
    const uint32_t scramble_mask = interleaved_mask;
    const uint32_t sample_hash = hash_sample_id ? murmur_hash3_mix(0, sample_id) : 0u;
    for (uint32_t ie = 2u * (uint32_t)find_msb(min_dim_mask + 1u); ie > 0u;) {
        uint32_t perm = murmur_hash3_finalize(murmur_hash3_mix(sample_hash, linear >> ie));
        const bool swap = (perm & 0x4u) != 0u;
        perm &= 0x3u;
        ie -= 2u;
        scrambled ^= (perm << ie) & scramble_mask;
        const uint32_t swap_mask = swap ? (0x3u << ie) : 0u;
        if (swap_mask == (scramble_mask & swap_mask)) scrambled ^= swap_vec & swap_mask;
    }
    if (hash_tile_id) scrambled &= pcount - 1u;
    return sample_id * pcount + scrambled;
}
// rendering/pointsets/sobol.glsl:74-108
static inline float sobol_point(const PointSetTable &t, uint32_t index, uint32_t dimension, uint32_t scramble) {
    dimension &= uint32_t(SobolData_Dimensions - 1);
    uint32_t result = scramble;
    for (uint32_t i = dimension * SobolData_MatrixSize; index != 0; index >>= 1, ++i)
        if (index & 1u) result ^= t.words[i];
    if (t.variant == RPTR_RNG_VARIANT_Z_SBL && dimension < 2) {
        const uint32_t tile_bits = (uint32_t)__builtin_popcount(SobolData_TileSize - 1);
        result ^= result << tile_bits;
    }
    return ldexpf((float)result, -32);
}
// sobol.glsl:112-130
static inline uint32_t sobol_shift_invert(const PointSetTable &t, uint32_t index, uint32_t index_shift) {
    index += index_shift;
    uint32_t r0 = 0, r1 = 0;
    for (uint32_t i = 0; index != 0; index >>= 1, ++i)
        if (index & 1u) {
            r0 ^= t.words[0 * SobolData_MatrixSize + i];
            r1 ^= t.words[1 * SobolData_MatrixSize + i];
        }
    const uint32_t tile_bits = (uint32_t)__builtin_popcount(SobolData_TileSize - 1);
    r0 >>= 32 - tile_bits;
    r1 >>= 32 - tile_bits;
    return index_shift + t.words[SobolData_Dimensions * SobolData_MatrixSize + r1 * SobolData_TileSize + r0];
}
// bn_rng.glsl:30-71 with BN_OPTIMIZED_DIMENSION_REPEAT and BN_OPTIMIZED_SPP 1
static inline float sample_bnd(const PointSetTable &t, uint32_t pixelID, uint32_t sampleID, uint32_t d) {
    const uint32_t T = BNData_TileSize, S = BNData_ScramblingDimensions;
    const uint32_t x_doffset = d / S;
    pixelID = ((pixelID + x_doffset) & (T - 1u)) + (pixelID & ~(T - 1u));
    d = (d & (S - 1u)) + x_doffset / T * S;
    d &= uint32_t(BNData_Dimensions - 1);
    if (sampleID & 1u) pixelID ^= T - 1u;
    if (sampleID & 2u) pixelID ^= (T - 1u) * T;
    const uint32_t x_soffset = sampleID * 73u, y_soffset = sampleID * 97u;
    pixelID = ((pixelID + x_soffset) & (T - 1u)) + (pixelID & ~(T - 1u));
    pixelID = ((pixelID + y_soffset * T) & (T * (T - 1u))) + (pixelID & ~(T * (T - 1u)));
    sampleID = 0;
    const uint32_t rankingIndex = pixelID * S + (d & (S - 1u));
    uint32_t value = t.words[d + sampleID * BNData_Dimensions];
    value ^= t.words[BNData_SampleCount * BNData_Dimensions + rankingIndex]; // tile_scrambling_yx_d_1spp
    return (0.5f + (float)value) / 256.0f;
}
// GET_RNG(index, frame, uvec4(pixel, dims)): lcg_rng.glsl:36-39 | sobol.glsl:165-195 | bn_rng.glsl:80-92,112 (which takes
// view_params.frame_id / frame_offset instead of the arguments)
static inline RandomState get_rng(const PointSetTable *table, uint32_t index, uint32_t frame, uint32_t px, uint32_t py, uint32_t dimx,
                                  uint32_t view_frame_id, uint32_t view_frame_offset) {
    RandomState r;
    r.table = (table && table->variant != RPTR_RNG_VARIANT_UNIFORM) ? table : nullptr;
    switch (r.variant()) {
    case RPTR_RNG_VARIANT_UNIFORM:
        r.lcg = get_lcg_rng(index, frame, px, py, dimx);
        break;
    case RPTR_RNG_VARIANT_BN:
        r.lcg.state = 0;
        r.pixel = (px & uint32_t(BNData_TileSize - 1)) + (py & uint32_t(BNData_TileSize - 1)) * BNData_TileSize;
        r.index = view_frame_id + view_frame_offset * 13u;
        break;
    default: {
        uint32_t linear = px + py * dimx;
        uint32_t sample_id = index;
        if (r.variant() == RPTR_RNG_VARIANT_Z_SBL) {
            const uint32_t T = SobolData_TileSize;
            const uint32_t sample_offset = morton_sample_id(0, px, py, T, T, true, false) & (T * T - 1u);
            sample_id = sobol_shift_invert(*table, sample_offset, T * T * sample_id);
            const uint32_t tile_bits = (uint32_t)__builtin_popcount(T - 1u);
            linear = (px >> tile_bits) + (py >> tile_bits) * (dimx >> tile_bits);
        }
        r.index = sample_id;
        r.lcg = get_lcg_rng(frame, 0, linear);
    }
    }
    r.dimension = 0;
    return r;
}
static inline float random_float1(RandomState &r, int dim) {
    switch (r.variant()) {
    case RPTR_RNG_VARIANT_UNIFORM: return lcg_randomf(r.lcg);
    case RPTR_RNG_VARIANT_BN: return sample_bnd(*r.table, r.pixel, r.index, uint32_t(r.dimension + dim));
    default: return sobol_point(*r.table, r.index, uint32_t(r.dimension) + uint32_t(dim), lcg_random(r.lcg));
    }
}
static inline vec2 random_float2(RandomState &r, int dim) { // defaults.glsl:29-35
    vec2 v;
    v.x = random_float1(r, dim);
    v.y = random_float1(r, dim + 1);
    return v;
}
static inline void random_shift_dim(RandomState &r, int dim_offset) { r.dimension += dim_offset; }
static inline void random_set_dim(RandomState &r, int dim) { r.dimension = dim; }
// rendering/pathspace.h
enum { DIM_PIXEL_X = 0, DIM_CAMERA_END = 6, DIM_DIRECTION_X = 0, DIM_LOBE = 2, DIM_FREE_PATH = 3, DIM_VERTEX_END = 4, DIM_RR = DIM_FREE_PATH - DIM_VERTEX_END,
       DIM_LIGHT_SEL_1 = 0, DIM_POSITION_X = 2, DIM_LIGHT_END = 4 };

// ---------------------------------------------------------------- util.glsl
// rendering/util.glsl:73-87
static inline void ortho_basis(vec3 &v_x, vec3 &v_y, const vec3 n) {
    v_y = vec3(0, 0, 0);
    if (n.x < 0.6f && n.x > -0.6f)
        v_y.x = 1.f;
    else if (n.y < 0.6f && n.y > -0.6f)
        v_y.y = 1.f;
    else if (n.z < 0.6f && n.z > -0.6f)
        v_y.z = 1.f;
    else
        v_y.x = 1.f;
    v_x = normalize(cross(v_y, n));
    v_y = normalize(cross(n, v_x));
}
// rendering/util.glsl:89-93
static inline mat3 ortho_frame(const vec3 n) {
    vec3 v_x, v_y;
    ortho_basis(v_x, v_y, n);
    return mat3(v_x, v_y, n);
}
// rendering/util.glsl:95-97
static inline float luminance(const vec3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }
static inline float pow2(float x) { return x * x; }
// integer power used where the reference writes pow(x, 5): evaluated as
// ((x*x)*(x*x))*x, the same on host and device (documented deviation from a
// libm pow that is within the Vulkan precision envelope of pow()).
static inline float pow5(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}
// rendering/util.glsl:120-122
static inline float cos_half_angle(float cos_angle) { return (1.0f + cos_angle) / sqrtf(2.0f + 2.0f * cos_angle); }
// rendering/util.glsl:151-153
static inline float mix_fma(float x, float y, float a) { return fmaf(a, y, fmaf(-a, x, x)); }
// rendering/util.glsl:19-28
static inline float positive_pow(float base, float power) { return powf(fmaxf(fabsf(base), 1.192092896e-07f), power); }
static inline float linear_to_srgb(float x) {
    return (x <= 0.0031308f) ? 12.92f * x : 1.055f * positive_pow(x, 1.f / 2.4f) - 0.055f;
}

// ---------------------------------------------------------------- dequantize
// librender/dequantize.glsl:8-21 (non-split 64-bit form)
static inline vec3 dequantize_position(uint64_t w, vec3 scaling, vec3 offset) {
    vec3 q((float)(uint32_t(w) & 0x1FFFFFu), (float)(uint32_t(w >> 21) & 0x1FFFFFu), (float)(uint32_t(w >> 42) & 0x1FFFFFu));
    return q * scaling + offset;
}
// librender/dequantize.glsl:23-41
static inline vec3 dequantize_normal(uint32_t word) {
    vec2 n = vec2((float)(int(word & 0xFFFF) - 0x8000), (float)(int(word >> 16) - 0x8000)) / float(0x7FFF);
    float nl1 = fabsf(n.x) + fabsf(n.y);
    if (nl1 >= 1.0f) {
        n = (vec2(1.0f) - vec2(fabsf(n.y), fabsf(n.x))) * vec2(n.x >= 0.0f ? 1.0f : -1.0f, n.y >= 0.0f ? 1.0f : -1.0f);
    }
    return normalize(vec3(n.x, n.y, 1.0f - nl1));
}
// librender/dequantize.glsl:43-48
static inline vec2 dequantize_uv(uint32_t word) {
    return vec2(0.0f, 1.0f) + vec2((float)int(word & 0xFFFF), (float)(-int(word >> 16))) * (8.0f / float(0xFFFFu));
}
// librender/quantize.h:7-11
static inline uint64_t quantize_position(vec3 p, vec3 extent, vec3 base) {
    p = (p - base) * float(0x200000u) / extent;
    // glm::uvec3(p): float -> unsigned conversion (truncation), then min
    uint32_t ux = (uint32_t)p.x, uy = (uint32_t)p.y, uz = (uint32_t)p.z;
    if (ux > 0x1FFFFFu) ux = 0x1FFFFFu;
    if (uy > 0x1FFFFFu) uy = 0x1FFFFFu;
    if (uz > 0x1FFFFFu) uz = 0x1FFFFFu;
    return uint64_t(ux) | (uint64_t(uy) << 21) | (uint64_t(uz) << 42);
}
// librender/quantize.h:13-18
static inline vec3 dequantization_scaling(vec3 extent) { return extent / float(0x200000u); }
static inline vec3 dequantization_offset(vec3 base, vec3 extent) { return base + extent * 0.5f / float(0x200000u); }
// librender/quantize.h:21-35
static inline uint32_t quantize_normal(vec3 n) {
    float nl1 = fabsf(n.x) + fabsf(n.y) + fabsf(n.z);
    vec2 pn = vec2(n.x, n.y) / nl1;
    if (n.z <= 0.0f)
        pn = (vec2(1.0f) - vec2(fabsf(pn.y), fabsf(pn.x))) * vec2(pn.x >= 0.0f ? 1.0f : -1.0f, pn.y >= 0.0f ? 1.0f : -1.0f);
    pn = pn * float(0x8000u);
    int ix = (int)pn.x, iy = (int)pn.y;
    ix = ix < -0x7FFF ? -0x7FFF : (ix > 0x7FFF ? 0x7FFF : ix);
    iy = iy < -0x7FFF ? -0x7FFF : (iy > 0x7FFF ? 0x7FFF : iy);
    uint32_t ux = uint32_t(0x8000 + ix), uy = uint32_t(0x8000 + iy);
    return ux | (uy << 16);
}
// librender/quantize.h:38-42 (safety_offset = 0)
static inline uint32_t quantize_uv(vec2 uv) {
    vec2 s = vec2(0.0f + uv.x, (1.0f + 0.0f) - uv.y) * (float(0xFFFFu) / 8.0f);
    uint32_t ux = uint32_t(0.5f + s.x) & 0xFFFFu, uy = uint32_t(0.5f + s.y) & 0xFFFFu;
    return ux | (uy << 16);
}

// ---------------------------------------------------------------- hit attributes
// rendering/rt/hit.glsl:12-23
struct RTHit {
    vec3 normal;
    float dist;
    vec3 geo_normal;
    int material_id;
    vec3 tangent;
    float bitangent_l;
    vec2 uv;
};
// rendering/rt/hit.glsl:49-56
static inline int calc_hit_material_id(int material_id_in, const uint8_t *ids, uint32_t primitive_id) {
    if (material_id_in < 0) {
        uint32_t material_id = ids[primitive_id]; // (id_4pack[p>>2] >> 8*(p&3)) & 0xff on little endian
        return int(material_id) - material_id_in - 1;
    }
    return material_id_in;
}
// rendering/rt/hit.glsl:58-128 (core) fed by the quantised overload :162-203
static inline RTHit calc_hit_attributes(float ray_t, uint32_t primitive_id, vec2 attrib, const mat3 &verts,
                                        const mat3 &normals_to_world, const mat3 &normals, bool has_normals,
                                        const mat3x2 &uvs, bool has_uvs, int material_id, const uint8_t *mat_ids) {
    RTHit payload;
    payload.dist = ray_t;
    vec3 gn = cross(verts[1] - verts[0], verts[2] - verts[0]);
    vec3 n = gn;
    const vec3 bary(1.f - attrib.x - attrib.y, attrib.x, attrib.y);
    if (has_normals) {
        n = normals * bary;
        if (dot(n, gn) < 0.0f)
            gn = -gn;
    }
    payload.geo_normal = gn * 0.5f;
    payload.normal = n;
    vec2 uv(0, 0);
    if (has_uvs)
        uv = uvs * bary;
    payload.uv = uv;
    payload.material_id = calc_hit_material_id(material_id, mat_ids, primitive_id);
    bool requires_tangent = true;
    payload.geo_normal = normals_to_world * payload.geo_normal;
    payload.normal = normalize(normals_to_world * payload.normal);
    if (requires_tangent && has_uvs) {
        float posframe_det = length(gn);
        vec3 frame_n = gn / (posframe_det * posframe_det);
        vec3 dp2perp = cross(verts[2] - verts[0], frame_n);
        vec3 dp1perp = cross(frame_n, verts[1] - verts[0]);
        vec2 duv1 = uvs.c[1] - uvs.c[0];
        vec2 duv2 = uvs.c[2] - uvs.c[0];
        vec3 T = dp2perp * duv1.x + dp1perp * duv2.x;
        vec3 B = dp2perp * duv1.y + dp1perp * duv2.y;
        T = normals_to_world * T;
        B = normals_to_world * B;
        float Tlen = length(T);
        if (Tlen > 0.0f && !std::isinf(Tlen) && !std::isnan(Tlen)) {
            payload.tangent = T;
            payload.bitangent_l = dot(normalize(cross(payload.geo_normal, T)), B);
            requires_tangent = false;
        }
    }
    if (requires_tangent) {
        payload.tangent = normalize(normals_to_world * cross(verts[2] - verts[0], gn));
        payload.bitangent_l = 1.0f;
    }
    return payload;
}

// ---------------------------------------------------------------- materials
// rendering/bsdfs/gltf_bsdf.glsl:15-35 (GLTF_SUPPORT_TRANSMISSION off: the
// PT_MEGAKERNEL build never defines MEGAKERNEL_MATERIALS)
struct GLTFMaterial {
    vec3 base_color;
    float metallic;
    float specular;
    float roughness;
    float ior;
    uint32_t flags;
};
// rendering/bsdfs/simple_bsdf.glsl:18-28
struct SimpleMaterial {
    vec3 base_color;
    float roughness;
    float ior;
    uint32_t flags;
};
// ---- textures: the reference's material sampler (render_vulkan.cpp:1657-1670: linear filter, linear mip filter, REPEAT addressing,
// LOD range 0..16, anisotropy 12) restated in software after the Vulkan specification's "Texel filtering" equations -- what the texture
// unit of the reference's GPU implements up to its fixed-point weights: "parity unpinned", tolerance-level agreement only.
// Texel centres at (i + 0.5) / size, bilinear weights in float, unorm byte / 255, sRGB decode (IEC 61966-2-1) per texel before filtering.
//   textureLod(uv, lod): lod clamped to the texture's levels, the two nearest levels blended by the fraction.
//   textureGrad(uv, ddx, ddy): rho_x = |ddx * size|, rho_y = |ddy * size|; eta = min(rho_max / rho_min, 12); N = ceil(eta) taps at
//   uv + major * (i / (N + 1) - 1/2), i = 1..N, along the larger derivative, each a textureLod at log2(rho_max / eta), averaged.
//   A footprint inside one texel (rho_max <= 1: magnification; zero derivatives: the any-hit alpha test, pt_megakernel.glsl:205) and
//   1 x 1 textures = one bilinear tap of level 0.
// Mip levels: RptrTextureDesc.mip_levels levels stored back to back, level l = max(1, w >> l) x max(1, h >> l)
// (vulkan/resource_utils.cpp:86-100); the reference uploads the levels its .vkt files hold and generates none.
struct TextureTable {
    const RptrTextureDesc *textures = nullptr;
    uint32_t num_textures = 0;
    float srgb_lut[256];
    TextureTable() {
        for (int i = 0; i < 256; ++i) {
            const float c = float(i) / 255.0f;
            srgb_lut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
        }
    }
};
// HitPoint::uv + HitPoint::duvdxy (rendering/rt/hit.glsl): where and over which footprint a material reads its textures
struct TexCoord {
    vec2 uv, ddx, ddy;
    TexCoord(vec2 uv_) : uv(uv_), ddx(0, 0), ddy(0, 0) {}
    TexCoord(vec2 uv_, vec2 ddx_, vec2 ddy_) : uv(uv_), ddx(ddx_), ddy(ddy_) {}
};
struct MipView {
    const uint8_t *texels;
    int w, h;
};
static inline int texture_levels(const RptrTextureDesc &t) { return t.mip_levels > 1u ? (int)t.mip_levels : 1; }
static inline MipView mip_view(const RptrTextureDesc &t, int level) {
    MipView v{t.rgba8, (int)t.width, (int)t.height};
    for (int l = 0; l < level; ++l) {
        v.texels += 4 * (size_t)v.w * (size_t)v.h;
        if (v.w > 1) v.w /= 2;
        if (v.h > 1) v.h /= 2;
    }
    return v;
}
static inline vec4 fetch_texel(const TextureTable &tt, const RptrTextureDesc &t, const MipView &v, int ix, int iy) {
    const uint8_t *c = v.texels + 4 * ((size_t)iy * v.w + (size_t)ix);
    if (t.srgb) return vec4(tt.srgb_lut[c[0]], tt.srgb_lut[c[1]], tt.srgb_lut[c[2]], float(c[3]) / 255.0f);
    return vec4(float(c[0]) / 255.0f, float(c[1]) / 255.0f, float(c[2]) / 255.0f, float(c[3]) / 255.0f);
}
static inline int wrap_repeat(int i, int n) {
    i %= n;
    return i < 0 ? i + n : i;
}
static inline vec4 texture_bilinear(const TextureTable &tt, const RptrTextureDesc &t, int level, vec2 uv) {
    const MipView v = mip_view(t, level);
    const int w = v.w, h = v.h;
    const float x = uv.x * float(w) - 0.5f, y = uv.y * float(h) - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int ix0 = wrap_repeat(int(x0), w), ix1 = wrap_repeat(int(x0) + 1, w);
    const int iy0 = wrap_repeat(int(y0), h), iy1 = wrap_repeat(int(y0) + 1, h);
    const vec4 c00 = fetch_texel(tt, t, v, ix0, iy0), c10 = fetch_texel(tt, t, v, ix1, iy0), c01 = fetch_texel(tt, t, v, ix0, iy1),
               c11 = fetch_texel(tt, t, v, ix1, iy1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const vec4 top(c00.x * gx + c10.x * fx, c00.y * gx + c10.y * fx, c00.z * gx + c10.z * fx, c00.w * gx + c10.w * fx);
    const vec4 bot(c01.x * gx + c11.x * fx, c01.y * gx + c11.y * fx, c01.z * gx + c11.z * fx, c01.w * gx + c11.w * fx);
    return vec4(top.x * gy + bot.x * fy, top.y * gy + bot.y * fy, top.z * gy + bot.z * fy, top.w * gy + bot.w * fy);
}
static inline vec4 texture_lod(const TextureTable &tt, int tex_id, vec2 uv, float lod) {
    const RptrTextureDesc &t = tt.textures[tex_id];
    const float last = float(texture_levels(t) - 1);
    lod = fminf(fmaxf(lod, 0.0f), fminf(last, 16.0f)); // (a NaN lod ends up at level 0)
    if (!(lod > 0.0f)) return texture_bilinear(tt, t, 0, uv);
    const float hi = floorf(lod), delta = lod - hi;
    const vec4 a = texture_bilinear(tt, t, int(hi), uv);
    if (delta == 0.0f) return a;
    const vec4 b = texture_bilinear(tt, t, int(hi) + 1, uv);
    const float g = 1.0f - delta;
    return vec4(a.x * g + b.x * delta, a.y * g + b.y * delta, a.z * g + b.z * delta, a.w * g + b.w * delta);
}
static inline vec4 texture_lod0(const TextureTable &tt, int tex_id, vec2 uv) { return texture_bilinear(tt, tt.textures[tex_id], 0, uv); }
#define ORC_MAX_ANISOTROPY 12.0f
static inline vec4 texture_grad(const TextureTable &tt, int tex_id, const TexCoord &tc) {
    const RptrTextureDesc &t = tt.textures[tex_id];
    const float w = float(t.width), h = float(t.height);
    const float mxx = tc.ddx.x * w, mxy = tc.ddx.y * h, myx = tc.ddy.x * w, myy = tc.ddy.y * h;
    const float rx = sqrtf(mxx * mxx + mxy * mxy), ry = sqrtf(myx * myx + myy * myy);
    const float rmax = fmaxf(rx, ry), rmin = fminf(rx, ry);
    // magnification (the footprint lies inside one texel), or nothing to filter (a 1 x 1 texture): one bilinear tap of level 0
    if (!(rmax > 1.0f) || (t.width == 1u && t.height == 1u)) return texture_bilinear(tt, t, 0, tc.uv);
    const float eta = rmin > 0.0f ? fminf(rmax / rmin, ORC_MAX_ANISOTROPY) : ORC_MAX_ANISOTROPY;
    const int n = (int)ceilf(eta);
    const float lod = log2f(rmax / eta);
    const vec2 major = rx > ry ? tc.ddx : tc.ddy;
    vec4 sum(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = 1; i <= n; ++i) {
        const float at = float(i) / float(n + 1) - 0.5f;
        const vec4 c = texture_lod(tt, tex_id, vec2(tc.uv.x + major.x * at, tc.uv.y + major.y * at), lod);
        sum = vec4(sum.x + c.x, sum.y + c.y, sum.z + c.z, sum.w + c.w);
    }
    const float inv = 1.0f / float(n);
    return vec4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
}
// rendering/rt/material_textures.glsl:37-60 (textureGrad: USE_MIPMAPPING is defined, librender/render_params.glsl.h:8)
static inline bool is_textured_param(float x) { return (float_bits(x) & RPTR_TEXTURED_PARAM_MASK) != 0u; }
static inline vec4 textured_color_param(const TextureTable &tt, vec4 x, const TexCoord &uv) {
    const uint32_t mask = float_bits(x.x);
    if (mask & RPTR_TEXTURED_PARAM_MASK) return texture_grad(tt, int(RPTR_TEXTURE_ID(mask)), uv);
    return x;
}
static inline float textured_scalar_param(const TextureTable &tt, float x, const TexCoord &uv) {
    const uint32_t mask = float_bits(x);
    if (mask & RPTR_TEXTURED_PARAM_MASK) {
        const vec4 t = texture_grad(tt, int(RPTR_TEXTURE_ID(mask)), uv);
        const uint32_t ch = RPTR_TEXTURE_CHANNEL(mask);
        return ch == 0 ? t.x : ch == 1 ? t.y : ch == 2 ? t.z : t.w;
    }
    return x;
}
// rendering/rt/material_textures.glsl:95-135, non-unrolled standard textures (a parameter is a literal or a texture handle);
// PREMULTIPLIED_BASE_COLOR_ALPHA is defined (vulkan/gpu_params.glsl:12)
static inline float unpack_material(const TextureTable &tt, GLTFMaterial &mat, vec3 &emitter_radiance, const RptrBaseMaterial &p, const TexCoord &uv) {
    const vec4 texel = textured_color_param(tt, vec4(p.base_color[0], p.base_color[1], p.base_color[2], 1.0f), uv);
    float alpha = texel.w;
    mat.base_color = vec3(texel.x, texel.y, texel.z);
    if (alpha > 0.001f)
        mat.base_color /= alpha;
    mat.specular = textured_scalar_param(tt, p.specular, uv);
    mat.roughness = textured_scalar_param(tt, p.roughness, uv);
    mat.metallic = textured_scalar_param(tt, p.metallic, uv);
    mat.ior = textured_scalar_param(tt, p.ior, uv);
    emitter_radiance = vec3(p.base_color[0], p.base_color[1], p.base_color[2]) * p.emission_intensity;
    if (p.emission_intensity != 0.0f) {
        if (is_textured_param(p.base_color[0])) emitter_radiance = mat.base_color * p.emission_intensity;
        mat.base_color = vec3(0.0f);
    }
    mat.flags = p.flags; // load_material, gltf_bsdf.glsl:38-62
    return alpha;
}
// same with SIMPLIFIED_MATERIAL (simple_bsdf.glsl:14-16,31-39)
static inline float unpack_material(const TextureTable &tt, SimpleMaterial &mat, vec3 &emitter_radiance, const RptrBaseMaterial &p, const TexCoord &uv) {
    const vec4 texel = textured_color_param(tt, vec4(p.base_color[0], p.base_color[1], p.base_color[2], 1.0f), uv);
    float alpha = texel.w;
    mat.base_color = vec3(texel.x, texel.y, texel.z);
    if (alpha > 0.001f)
        mat.base_color /= alpha;
    emitter_radiance = vec3(p.base_color[0], p.base_color[1], p.base_color[2]) * p.emission_intensity;
    if (p.emission_intensity != 0.0f) {
        if (is_textured_param(p.base_color[0])) emitter_radiance = mat.base_color * p.emission_intensity;
        mat.base_color = vec3(0.0f);
    }
    mat.roughness = 1.0f;
    mat.ior = 1.0f;
    mat.flags = p.flags;
    return alpha;
}

// ---------------------------------------------------------------- glTF BSDF
// rendering/bsdfs/gltf_bsdf.glsl:172-174
static inline float schlick_weight(float local_cos_theta) { return pow5(clampf(1.f - local_cos_theta, 0.f, 1.f)); }
// :194-198
static inline float gtr_2(float cos_theta_h, float alpha) {
    float alpha_sqr = alpha * alpha;
    return M_1_PIf * alpha_sqr / pow2(1.f + (alpha_sqr - 1.f) * cos_theta_h * cos_theta_h);
}
// :200-202
static inline float smith_visibility_den1(float n_dot_o, float alpha_sq) {
    return fabsf(n_dot_o) + sqrtf(alpha_sq + (1.0f - alpha_sq) * n_dot_o * n_dot_o);
}
// :207-212
static inline float smith_visibility_ggx(float n_dot_o, float n_dot_i, float alpha_g) {
    float a = alpha_g * alpha_g;
    float den_shad = smith_visibility_den1(n_dot_i, a);
    float den_mask = smith_visibility_den1(n_dot_o, a);
    return 1.f / (den_shad * den_mask);
}
// :216-222
static inline vec3 to_pipe_sample(const vec2 U) {
    float phi = 2.0f * M_PIf * U.x;
    return vec3(cosf(phi), sinf(phi), U.y);
}
// :225-229
static inline vec3 sample_sphere(vec3 UP) {
    float cos_theta = UP.z * 2.0f - 1.0f;
    float sin_theta = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    return vec3(sin_theta * UP.x, sin_theta * UP.y, cos_theta);
}
// :233-250
static inline vec3 sample_gtr_2_vndf(const vec3 w_o_local, vec2 alpha, const vec3 UP) {
    vec3 wiStd = normalize(vec3(alpha.x * w_o_local.x, alpha.y * w_o_local.y, w_o_local.z));
    float z = fmaf((1.0f - UP.z), (1.0f + wiStd.z), -wiStd.z);
    float sinTheta = sqrtf(clampf(1.0f - z * z, 0.0f, 1.0f));
    float x = sinTheta * UP.x;
    float y = sinTheta * UP.y;
    vec3 wmStd = vec3(x, y, z) + wiStd;
    vec3 wm = vec3(wmStd.x * alpha.x, wmStd.y * alpha.y, fmaxf(0.0f, wmStd.z));
    float wmL = length(wm);
    return wm / wmL;
}
// :253-257
static inline float gtr_2_vndf_pdf(float n_dot_o, float cos_theta_h, float alpha) {
    return gtr_2(cos_theta_h, alpha) * (0.5f / smith_visibility_den1(n_dot_o, alpha * alpha));
}
// :259-261
static inline vec3 gltf_diffuse_basecolor(const GLTFMaterial &mat) { return (1.0f - mat.metallic) * mat.base_color; }
// :263-273 (GLTF_SUPPORT_TINT off)
static inline vec3 gltf_specular_basecolor(const GLTFMaterial &mat, float ior) {
    vec3 dielectric_base = vec3(pow2((ior - 1.0f) / (ior + 1.0f)));
    return mix(dielectric_base, mat.base_color, mat.metallic);
}
// :275-277
static inline float gltf_specular_alpha(const GLTFMaterial &mat) { return fmaxf(mat.roughness * mat.roughness, 0.002f); }
// :284-292
static inline float gltf_schlick_weight(float local_o_dot_h, float ior) {
    float f_weight = schlick_weight(local_o_dot_h);
    if (ior < 1.0f) {
        float cos_critical = sqrtf(1.0f - ior * ior);
        f_weight = mix(f_weight, 1.0f, fminf((1.0f - local_o_dot_h) / (1.0f - cos_critical), 1.0f));
    }
    return f_weight;
}
// :294-359 (no transmission)
static inline vec3 gltf_bsdf(const GLTFMaterial &mat, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    float ior = o_dot_n < 0.0f ? 1.0f / mat.ior : mat.ior;
    vec3 w_h;
    if (i_dot_n * o_dot_n < 0.0f)
        return vec3(0.0f);
    else
        w_h = w_i + w_o;
    w_h = normalize(w_h);
    float o_dot_h = dot(w_o, w_h);
    vec3 diffuse = gltf_diffuse_basecolor(mat) * float(M_1_PIf);
    vec3 specular = vec3(0.0f);
    if (mat.ior > 1.0f) {
        vec3 f0 = gltf_specular_basecolor(mat, mat.ior);
        float specular_alpha = gltf_specular_alpha(mat);
        float specular_refl = gtr_2(dot(n, w_h), specular_alpha);
        specular_refl *= smith_visibility_ggx(o_dot_n, i_dot_n, specular_alpha);
        float f_weight = gltf_schlick_weight(fabsf(o_dot_h), ior);
        vec3 F = mix(f0, vec3(1.0f), f_weight);
        diffuse *= (vec3(1.0f) - F);
        specular = specular_refl * F;
    }
    return diffuse + specular;
}
// :366-394 (GLTF_COMPONENT_COUNT = 2)
struct GLTFComponentSampler {
    float weights[2];
};
static inline GLTFComponentSampler gltf_component_sampler(const GLTFMaterial &mat, float ior, vec4 o_dot_h, vec4 visibility) {
    GLTFComponentSampler components;
    float specular_base_lum = luminance(gltf_specular_basecolor(mat, mat.ior));
    float F0 = mix(specular_base_lum, 1.0f, gltf_schlick_weight(o_dot_h.x, 1.0f));
    float F1 = mix(specular_base_lum, 1.0f, gltf_schlick_weight(o_dot_h.y, 1.0f));
    (void)ior; // F2 only feeds the transmission component
    components.weights[0] = (1.0f - F0) * visibility.x * (1.0f - mat.metallic) * luminance(gltf_diffuse_basecolor(mat));
    components.weights[1] = F1 * visibility.y;
    float weight_sum = 0.0f;
    for (int i = 0; i < 2; ++i)
        weight_sum += components.weights[i];
    if (weight_sum > 0.0f) {
        for (int i = 0; i < 2; ++i)
            components.weights[i] /= weight_sum;
    } else
        components.weights[0] = 1.0f;
    return components;
}
// :395-409
static inline int glft_sample_reuse_component(const GLTFComponentSampler &components, float &rnd, float &component_probability) {
    int component = 0;
    float next_layer_p_base = 0.0f, layer_p_base = 0.0f;
    for (int i = 0; i < 2; ++i) {
        float layer_p = components.weights[i];
        if (layer_p > 0.0f && rnd >= next_layer_p_base) {
            component = i;
            component_probability = layer_p;
            layer_p_base = next_layer_p_base;
        }
        next_layer_p_base += layer_p;
    }
    rnd = fminf(1.0f, (rnd - layer_p_base) / component_probability);
    return component;
}
// :414-494 (no transmission)
static inline float gltf_wpdf(const GLTFMaterial &mat, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    float ior = o_dot_n < 0.0f ? 1.0f / mat.ior : mat.ior;
    float pdf = M_1_PIf * fabsf(i_dot_n);
    if (mat.ior > 1.0f) {
        vec3 w_h;
        if (i_dot_n * o_dot_n < 0.0f)
            return 0.0f;
        else
            w_h = w_i + w_o;
        w_h = normalize(w_h);
        float o_dot_h = dot(w_o, w_h);
        float cos_theta_h = dot(w_h, n);
        vec4 visibility = vec4(0.0f);
        visibility.x = 1.0f;
        float specular_alpha = gltf_specular_alpha(mat);
        visibility.y = 2.0f * fabsf(i_dot_n) / smith_visibility_den1(i_dot_n, specular_alpha * specular_alpha);
        GLTFComponentSampler components = gltf_component_sampler(mat, ior, vec4(fabsf(o_dot_h)), visibility);
        float specular = gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        pdf *= components.weights[0];
        pdf += specular * components.weights[1];
    }
    return pdf;
}
// :496-645 (no transmission)
static inline vec3 sample_gltf_brdf(const GLTFMaterial &mat, const vec3 n, const vec3 w_o, vec3 &w_i, float &pdf,
                                    float &mis_wpdf, vec2 rng_sample, vec2 fresnel_sample, const vec3 v_x, const vec3 v_y) {
    vec3 w_o_local = transpose(mat3(v_x, v_y, n)) * w_o;
    float o_dot_n = w_o_local.z;
    float ior = mat.ior;
    if (o_dot_n < 0.0f) {
        pdf = 0.0f;
        return vec3(0.0f);
    }
    vec3 UP = to_pipe_sample(rng_sample);
    vec3 w_i_diffuse = normalize(n + sample_sphere(UP));
    float specular_alpha = gltf_specular_alpha(mat);
    int component = 0;
    float component_selection_pdf = 0.0f;
    GLTFComponentSampler components;
    components.weights[0] = components.weights[1] = 0.0f;
    vec3 w_h_specular_local;
    if (mat.ior > 1.0f) {
        vec4 o_dot_h_all = vec4(0.0f);
        vec4 visibility_all = vec4(0.0f);
        o_dot_h_all.x = cos_half_angle(dot(w_o, w_i_diffuse));
        visibility_all.x = 1.0f;
        w_h_specular_local = sample_gtr_2_vndf(w_o_local, vec2(specular_alpha), UP);
        o_dot_h_all.y = dot(w_o_local, w_h_specular_local);
        float spec_i_dot_n_local = reflect(-w_o_local, w_h_specular_local).z;
        visibility_all.y = spec_i_dot_n_local > 0.0f
                               ? 2.0f * spec_i_dot_n_local / smith_visibility_den1(spec_i_dot_n_local, specular_alpha * specular_alpha)
                               : 0.0f;
        components = gltf_component_sampler(mat, ior, o_dot_h_all, visibility_all);
        component = glft_sample_reuse_component(components, fresnel_sample.x, component_selection_pdf);
    }
    float cos_theta_h;
    if (component == 0) {
        w_i = w_i_diffuse;
        vec3 w_h = normalize(w_i + w_o);
        cos_theta_h = dot(n, w_h);
    } else {
        vec3 w_h = w_h_specular_local;
        cos_theta_h = w_h.z;
        w_h = mat3(v_x, v_y, n) * w_h;
        w_i = reflect(-w_o, w_h);
    }
    float i_dot_n = dot(n, w_i);
    if (!(i_dot_n > 0.0f)) {
        pdf = 0.0f;
        return vec3(0.0f);
    }
    float diffuse = M_1_PIf * fabsf(i_dot_n);
    pdf = diffuse;
    if (mat.ior > 1.0f) {
        pdf *= components.weights[0];
        float specular = gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        pdf += specular * components.weights[1];
    }
    if (!(pdf > 0.0f))
        return vec3(0.0f);
    vec3 result = gltf_bsdf(mat, n, w_o, w_i);
    mis_wpdf = gltf_wpdf(mat, n, w_o, w_i);
    return result * fabsf(i_dot_n) / pdf;
}

// ---------------------------------------------------------------- simple BSDF
// rendering/bsdfs/simple_bsdf.glsl:44-59
static inline vec3 simple_bsdf(const SimpleMaterial &mat, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    vec3 diffuse = mat.base_color * float(M_1_PIf);
    if (i_dot_n * o_dot_n < 0.0f)
        return vec3(0.0f);
    return diffuse;
}
// :61-66
static inline vec3 simple_sample_sphere(vec2 rnd) {
    float phi = 2.0f * M_PIf * rnd.x;
    float cos_theta = rnd.y * 2.0f - 1.0f;
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return vec3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
}
// :68-83
static inline float simple_pdf(const SimpleMaterial &, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    float pdf = M_1_PIf * fabsf(i_dot_n);
    if (i_dot_n * o_dot_n < 0.0f)
        return 0.0f;
    return pdf;
}
// :85-94
static inline vec3 sample_simple_brdf(const SimpleMaterial &mat, const vec3 n, const vec3, vec3 &w_i, float &pdf,
                                      float &mis_pdf, vec2 rng_sample) {
    w_i = normalize(n + simple_sample_sphere(rng_sample));
    float i_dot_n = dot(n, w_i);
    pdf = mis_pdf = M_1_PIf * fabsf(i_dot_n);
    return mat.base_color;
}

// ---------------------------------------------------------------- glTF BSDF with the transmission lobe
// rendering/bsdfs/gltf_bsdf.glsl compiled with GLTF_SUPPORT_TRANSMISSION + GLTF_SUPPORT_TRANSMISSION_ROUGHNESS (what
// MEGAKERNEL_MATERIALS switches on, :10-13; the RT-pipeline hit groups of vulkan/CMakeLists.txt:35-40 build it). Three components:
// diffuse, GGX reflection, GGX transmission -- through a ONESIDED surface a refraction (w_h = -ior w_i - w_o, angle compression),
// through a two-sided one the "thin" double reflection w_i = reflect(reflect(-w_o, w_h), n).
struct GLTFTransMaterial { // :15-35
    vec3 base_color;
    float metallic;
    float specular;
    float roughness;
    float ior;
    float transmission_roughness;
    float specular_transmission;
    vec3 transmission_color;
    uint32_t flags;
};
// GLSL refract(I, N, eta)
static inline vec3 refract(vec3 I, vec3 N, float eta) {
    const float d = dot(N, I);
    const float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return vec3(0.0f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}
// material_textures.glsl:95-135 + load_material gltf_bsdf.glsl:38-62
static inline float unpack_material(const TextureTable &tt, GLTFTransMaterial &mat, vec3 &emitter_radiance, const RptrBaseMaterial &p, const TexCoord &uv) {
    GLTFMaterial base;
    float alpha = unpack_material(tt, base, emitter_radiance, p, uv);
    mat.base_color = base.base_color;
    mat.metallic = base.metallic;
    mat.specular = base.specular;
    mat.roughness = base.roughness;
    mat.ior = base.ior;
    mat.flags = base.flags;
    mat.transmission_roughness = 0.0f;
    mat.transmission_color = vec3(0.0f);
    mat.specular_transmission = textured_scalar_param(tt, p.specular_transmission, uv);
    if (mat.specular_transmission > 0.0f) {
        if (!(mat.ior > 1.0f)) {
            alpha *= 1.0f - mat.specular_transmission;
            mat.specular_transmission = 0.0f;
        } else {
            mat.transmission_color = mat.base_color;
            mat.transmission_roughness = mat.roughness;
            mat.roughness = sqrtf(textured_scalar_param(tt, p.clearcoat_gloss, uv));
        }
    } else
        mat.transmission_color = vec3(0.0f);
    return alpha;
}
static inline vec3 gltf_diffuse_basecolor(const GLTFTransMaterial &mat) { return (1.0f - mat.metallic) * mat.base_color; }
static inline vec3 gltf_specular_basecolor(const GLTFTransMaterial &mat, float ior) {
    vec3 dielectric_base = vec3(pow2((ior - 1.0f) / (ior + 1.0f)));
    return mix(dielectric_base, mat.base_color, mat.metallic);
}
static inline float gltf_specular_alpha(const GLTFTransMaterial &mat) { return fmaxf(mat.roughness * mat.roughness, 0.002f); }
static inline float gltf_transmission_alpha(const GLTFTransMaterial &mat) { return fmaxf(mat.transmission_roughness * mat.transmission_roughness, 0.002f); } // :278-282
// :294-359
static inline vec3 gltf_bsdf(const GLTFTransMaterial &mat, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    float ior = o_dot_n < 0.0f ? 1.0f / mat.ior : mat.ior;
    vec3 w_h;
    if (i_dot_n * o_dot_n < 0.0f) {
        if (!(mat.specular_transmission > 0.f))
            return vec3(0.0f);
        if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0)
            w_h = -ior * w_i - w_o;
        else
            w_h = reflect(w_i, n) + w_o;
        if (!(dot(w_h, n) > 0.0f))
            return vec3(0.0f);
    } else
        w_h = w_i + w_o;
    w_h = normalize(w_h);
    float o_dot_h = dot(w_o, w_h), i_dot_h = dot(w_i, w_h);
    vec3 diffuse = gltf_diffuse_basecolor(mat) * float(M_1_PIf);
    vec3 specular = vec3(0.0f);
    if (mat.ior > 1.0f) {
        vec3 f0 = gltf_specular_basecolor(mat, mat.ior);
        float specular_alpha = gltf_specular_alpha(mat);
        if (i_dot_n * o_dot_n < 0.0f)
            specular_alpha = gltf_transmission_alpha(mat);
        float specular_refl = gtr_2(dot(n, w_h), specular_alpha);
        specular_refl *= smith_visibility_ggx(o_dot_n, i_dot_n, specular_alpha);
        float f_weight = gltf_schlick_weight(fabsf(o_dot_h), ior);
        vec3 F = mix(f0, vec3(1.0f), f_weight);
        if (i_dot_n * o_dot_n < 0.0f) {
            diffuse = vec3(0.0f);
            specular = ((specular_refl * (1.f - mat.metallic)) * mat.specular_transmission) * mat.transmission_color * (vec3(1.0f) - F);
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0) { // transmission angle compression
                float angle_compression = 2.0f * o_dot_h / (i_dot_h * ior + o_dot_h);
                specular *= angle_compression * angle_compression;
            }
        } else {
            diffuse *= (1.0f - mat.specular_transmission);
            diffuse *= (vec3(1.0f) - F);
            specular = specular_refl * F;
        }
    }
    return diffuse + specular;
}
// :366-394 (GLTF_COMPONENT_COUNT = 3)
struct GLTFComponentSampler3 {
    float weights[3];
};
static inline GLTFComponentSampler3 gltf_component_sampler(const GLTFTransMaterial &mat, float ior, vec4 o_dot_h, vec4 visibility) {
    GLTFComponentSampler3 components;
    float specular_base_lum = luminance(gltf_specular_basecolor(mat, mat.ior));
    float F0 = mix(specular_base_lum, 1.0f, gltf_schlick_weight(o_dot_h.x, 1.0f));
    float F1 = mix(specular_base_lum, 1.0f, gltf_schlick_weight(o_dot_h.y, 1.0f));
    float F2 = mix(specular_base_lum, 1.0f, gltf_schlick_weight(o_dot_h.z, ior));
    components.weights[0] = (1.0f - F0) * visibility.x * (1.0f - mat.metallic) * luminance(gltf_diffuse_basecolor(mat));
    components.weights[1] = F1 * visibility.y;
    components.weights[0] *= (1.0f - mat.specular_transmission);
    components.weights[2] = (1.0f - F2) * visibility.z * (1.0f - mat.metallic) * mat.specular_transmission;
    float weight_sum = 0.0f;
    for (int i = 0; i < 3; ++i)
        weight_sum += components.weights[i];
    if (weight_sum > 0.0f) {
        for (int i = 0; i < 3; ++i)
            components.weights[i] /= weight_sum;
    } else
        components.weights[0] = 1.0f;
    return components;
}
static inline int glft_sample_reuse_component(const GLTFComponentSampler3 &components, float &rnd, float &component_probability) { // :395-409
    int component = 0;
    float next_layer_p_base = 0.0f, layer_p_base = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float layer_p = components.weights[i];
        if (layer_p > 0.0f && rnd >= next_layer_p_base) {
            component = i;
            component_probability = layer_p;
            layer_p_base = next_layer_p_base;
        }
        next_layer_p_base += layer_p;
    }
    rnd = fminf(1.0f, (rnd - layer_p_base) / component_probability);
    return component;
}
// :414-494
static inline float gltf_wpdf(const GLTFTransMaterial &mat, const vec3 n, const vec3 w_o, const vec3 w_i) {
    float i_dot_n = dot(n, w_i);
    float o_dot_n = dot(n, w_o);
    float ior = o_dot_n < 0.0f ? 1.0f / mat.ior : mat.ior;
    float pdf = M_1_PIf * fabsf(i_dot_n);
    if (mat.ior > 1.0f) {
        vec3 w_h;
        if (i_dot_n * o_dot_n < 0.0f) {
            if (!(mat.specular_transmission > 0.f))
                return 0.0f;
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0)
                w_h = -ior * w_i - w_o;
            else
                w_h = reflect(w_i, n) + w_o;
            if (!(dot(w_h, n) > 0.0f))
                return 0.0f;
        } else
            w_h = w_i + w_o;
        w_h = normalize(w_h);
        float o_dot_h = dot(w_o, w_h), i_dot_h = dot(w_i, w_h);
        float cos_theta_h = dot(w_h, n);
        vec4 visibility = vec4(0.0f);
        visibility.x = 1.0f;
        float specular_alpha = gltf_specular_alpha(mat);
        visibility.y = 2.0f * fabsf(i_dot_n) / smith_visibility_den1(i_dot_n, specular_alpha * specular_alpha);
        visibility.z = visibility.y;
        float transmission_alpha = specular_alpha;
        if (mat.specular_transmission > 0.f) {
            transmission_alpha = gltf_transmission_alpha(mat);
            visibility.z = 2.0f * fabsf(i_dot_n) / smith_visibility_den1(i_dot_n, transmission_alpha * transmission_alpha);
        }
        GLTFComponentSampler3 components = gltf_component_sampler(mat, ior, vec4(fabsf(o_dot_h)), visibility);
        if (i_dot_n * o_dot_n < 0.0f)
            specular_alpha = transmission_alpha;
        float specular = gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        if (i_dot_n * o_dot_n < 0.0f) {
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0) {
                float angle_compression = 2.0f * o_dot_h / (i_dot_h * ior + o_dot_h);
                specular *= angle_compression * angle_compression;
            }
            pdf = specular * components.weights[2];
        } else {
            pdf *= components.weights[0];
            pdf += specular * components.weights[1];
        }
    }
    return pdf;
}
// :496-645
static inline vec3 sample_gltf_brdf(const GLTFTransMaterial &mat, const vec3 n, const vec3 w_o, vec3 &w_i, float &pdf,
                                    float &mis_wpdf, vec2 rng_sample, vec2 fresnel_sample, const vec3 v_x, const vec3 v_y) {
    vec3 w_o_local = transpose(mat3(v_x, v_y, n)) * w_o;
    float o_dot_n = w_o_local.z;
    float ior = o_dot_n < 0.0f ? 1.0f / mat.ior : mat.ior;
    if (o_dot_n < 0.0f)
        w_o_local.z = -w_o_local.z;
    vec3 UP = to_pipe_sample(rng_sample);
    vec3 w_i_diffuse = normalize(n + sample_sphere(UP));
    if (o_dot_n < 0.0f)
        w_i_diffuse = -w_i_diffuse;
    float specular_alpha = gltf_specular_alpha(mat);
    int component = 0;
    float component_selection_pdf = 0.0f;
    GLTFComponentSampler3 components;
    components.weights[0] = components.weights[1] = components.weights[2] = 0.0f;
    vec3 w_h_specular_local;
    vec3 w_h_transmission_local;
    if (mat.ior > 1.0f) {
        vec4 o_dot_h_all = vec4(0.0f);
        vec4 visibility_all = vec4(0.0f);
        o_dot_h_all.x = cos_half_angle(dot(w_o, w_i_diffuse));
        visibility_all.x = 1.0f;
        w_h_specular_local = sample_gtr_2_vndf(w_o_local, vec2(specular_alpha), UP);
        o_dot_h_all.y = dot(w_o_local, w_h_specular_local);
        float spec_i_dot_n_local = reflect(-w_o_local, w_h_specular_local).z;
        visibility_all.y = spec_i_dot_n_local > 0.0f
                               ? 2.0f * spec_i_dot_n_local / smith_visibility_den1(spec_i_dot_n_local, specular_alpha * specular_alpha)
                               : 0.0f;
        float transmission_alpha = specular_alpha;
        w_h_transmission_local = w_h_specular_local;
        o_dot_h_all.z = o_dot_h_all.y;
        float trans_i_dot_n_local = spec_i_dot_n_local;
        if (mat.specular_transmission > 0.f) {
            transmission_alpha = gltf_transmission_alpha(mat);
            w_h_transmission_local = sample_gtr_2_vndf(w_o_local, vec2(transmission_alpha), UP);
            o_dot_h_all.z = dot(w_o_local, w_h_transmission_local);
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0)
                trans_i_dot_n_local = -refract(-w_o_local, w_h_transmission_local, 1.0f / ior).z;
            else
                trans_i_dot_n_local = reflect(-w_o_local, w_h_transmission_local).z;
            visibility_all.z = trans_i_dot_n_local > 0.0f
                                   ? 2.0f * trans_i_dot_n_local / smith_visibility_den1(trans_i_dot_n_local, transmission_alpha * transmission_alpha)
                                   : 0.0f;
        }
        components = gltf_component_sampler(mat, ior, o_dot_h_all, visibility_all);
        component = glft_sample_reuse_component(components, fresnel_sample.x, component_selection_pdf);
    }
    float cos_theta_h;
    float i_dot_h;
    float o_dot_h;
    if (component == 0) {
        w_i = w_i_diffuse;
        vec3 w_h = normalize(w_i + w_o);
        cos_theta_h = dot(n, w_h);
        i_dot_h = o_dot_h = dot(w_o, w_h);
    } else {
        if (component == 2) {
            specular_alpha = gltf_transmission_alpha(mat);
            w_h_specular_local = w_h_transmission_local;
        }
        vec3 w_h = w_h_specular_local;
        if (o_dot_n < 0.0f)
            w_h.z = -w_h.z;
        cos_theta_h = w_h.z;
        w_h = mat3(v_x, v_y, n) * w_h;
        i_dot_h = o_dot_h = dot(w_o, w_h);
        if (component != 1) {
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0) {
                w_i = refract(-w_o, w_h, 1.0f / ior);
                i_dot_h = dot(w_i, w_h);
            } else
                w_i = reflect(reflect(-w_o, w_h), n);
        } else
            w_i = reflect(-w_o, w_h);
    }
    float i_dot_n = dot(n, w_i);
    if ((i_dot_n * o_dot_n > 0.0f) != (component != 2)) {
        pdf = 0.0f;
        return vec3(0.0f);
    }
    float diffuse = M_1_PIf * fabsf(i_dot_n);
    pdf = diffuse;
    if (mat.ior > 1.0f) {
        pdf *= components.weights[0];
        float specular = gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        if (i_dot_n * o_dot_n < 0.0f) {
            if ((mat.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0) {
                float angle_compression = 2.0f * o_dot_h / (i_dot_h * ior + o_dot_h);
                specular *= angle_compression * angle_compression;
            }
            pdf = specular * components.weights[2];
        } else
            pdf += specular * components.weights[1];
    }
    if (!(pdf > 0.0f))
        return vec3(0.0f);
    vec3 result = gltf_bsdf(mat, n, w_o, w_i);
    mis_wpdf = gltf_wpdf(mat, n, w_o, w_i);
    return result * fabsf(i_dot_n) / pdf;
}

// material registration (gltf_bsdf.glsl:649-655, simple_bsdf.glsl:98-104)
struct InteractionPoint { // rendering/bsdfs/hit_point.glsl:14-21
    vec3 p, gn, n, v_x, v_y;
    int primitiveId, instanceId;
};
static inline vec3 eval_bsdf(const GLTFMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return gltf_bsdf(m, h.n, w_o, w_i); }
static inline float eval_bsdf_wpdf(const GLTFMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return gltf_wpdf(m, h.n, w_o, w_i); }
static inline vec3 sample_bsdf(const GLTFMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 &w_i, float &pdf, float &mis,
                               vec2 rn_dir, vec2 rn_lobe) {
    return sample_gltf_brdf(m, h.n, w_o, w_i, pdf, mis, rn_dir, rn_lobe, h.v_x, h.v_y);
}
static inline vec3 eval_bsdf(const GLTFTransMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return gltf_bsdf(m, h.n, w_o, w_i); }
static inline float eval_bsdf_wpdf(const GLTFTransMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return gltf_wpdf(m, h.n, w_o, w_i); }
static inline vec3 sample_bsdf(const GLTFTransMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 &w_i, float &pdf, float &mis,
                               vec2 rn_dir, vec2 rn_lobe) {
    return sample_gltf_brdf(m, h.n, w_o, w_i, pdf, mis, rn_dir, rn_lobe, h.v_x, h.v_y);
}
static inline vec3 eval_bsdf(const SimpleMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return simple_bsdf(m, h.n, w_o, w_i); }
static inline float eval_bsdf_wpdf(const SimpleMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 w_i) { return simple_pdf(m, h.n, w_o, w_i); }
static inline vec3 sample_bsdf(const SimpleMaterial &m, const InteractionPoint &h, vec3 w_o, vec3 &w_i, float &pdf, float &mis,
                               vec2 rn_dir, vec2) {
    return sample_simple_brdf(m, h.n, w_o, w_i, pdf, mis, rn_dir);
}

// ---------------------------------------------------------------- triangle lights
struct TriLight { // rendering/lights/tri.h.glsl:8-11
    vec3 v0, v1, v2, radiance;
};
// rendering/lights/tri.glsl:12-19
static inline TriLight decode_tri_light(const RptrTriLightData &d) {
    TriLight l;
    l.v0 = vec3(d.v0[0], d.v0[1], d.v0[2]);
    l.v1 = vec3(d.v1[0], d.v1[1], d.v1[2]);
    l.v2 = vec3(d.v2[0], d.v2[1], d.v2[2]);
    l.radiance = vec3(d.radiance[0], d.radiance[1], d.radiance[2]);
    return l;
}
// tri.glsl:21-23
static inline bool is_tri_facing_forward(vec3 v0, vec3 v1, vec3 v2) { return dot(cross(v0, v1), v2) < 0.0f; }
// tri.glsl:58-74
static inline float fast_positive_atan(float y) {
    float rx, ry, rz;
    rx = (fabsf(y) > 1.0f) ? (1.0f / fabsf(y)) : fabsf(y);
    ry = rx * rx;
    rz = fmaf(ry, 0.02083509974181652f, -0.08513300120830536f);
    rz = fmaf(ry, rz, 0.18014100193977356f);
    rz = fmaf(ry, rz, -0.3302994966506958f);
    ry = fmaf(ry, rz, 0.9998660087585449f);
    rz = fmaf(-2.0f * ry, rx, float(0.5f * M_PIf));
    rz = (fabsf(y) > 1.0f) ? rz : 0.0f;
    rx = fmaf(rx, ry, rz);
    return (y < 0.0f) ? (M_PIf - rx) : rx;
}
// tri.glsl:83-113
static inline float half_triangle_solid_angle_tan(vec3 v0, vec3 v1, vec3 v2, vec3 &triangle_parameters) {
    float householder_sign = (v0.x > 0.0f) ? -1.0f : 1.0f;
    vec2 householder_yz = vec2(v0.y, v0.z) * (1.0f / (fabsf(v0.x) + 1.0f));
    float dot_0_1 = dot(v0, v1);
    float dot_0_2 = dot(v1, v2);
    float dot_1_2 = dot(v0, v2);
    float dot_householder_0 = fmaf(-householder_sign, v1.x, dot_0_1);
    float dot_householder_2 = fmaf(-householder_sign, v2.x, dot_1_2);
    mat2 bottom_right_minor(vec2(fmaf(-dot_householder_0, householder_yz.x, v1.y), fmaf(-dot_householder_0, householder_yz.y, v1.z)),
                            vec2(fmaf(-dot_householder_2, householder_yz.x, v2.y), fmaf(-dot_householder_2, householder_yz.y, v2.z)));
    float simplex_volume = fabsf(determinant(bottom_right_minor));
    float dot_0_2_plus_1_2 = dot_0_2 + dot_1_2;
    float one_plus_dot_0_1 = 1.0f + dot_0_1;
    float tangent = simplex_volume / (one_plus_dot_0_1 + dot_0_2_plus_1_2);
    triangle_parameters = vec3(simplex_volume, dot_0_2_plus_1_2, one_plus_dot_0_1);
    return tangent;
}
// tri.glsl:115-124
static inline float triangle_solid_angle(vec3 v0, vec3 v1, vec3 v2, vec3 &triangle_parameters) {
    return 2.0f * fast_positive_atan(half_triangle_solid_angle_tan(v0, v1, v2, triangle_parameters));
}
static inline float approx_triangle_solid_angle(vec3 v0, vec3 v1, vec3 v2) {
    vec3 tp;
    return 2.0f * fast_positive_atan(half_triangle_solid_angle_tan(v0, v1, v2, tp));
}
// tri.glsl:132-152
static inline vec3 sample_solid_angle_polygon(vec3 v0, vec3 v1, vec3 v2, float polygon_solid_angle, vec3 solid_angle_parameters,
                                              vec2 random_numbers) {
    float target_solid_angle = polygon_solid_angle * random_numbers[0];
    float subtriangle_solid_angle = target_solid_angle;
    vec3 parameters = solid_angle_parameters;
    vec3 vertices[3] = {v1, v0, v2};
    vec2 cos_sin = vec2(cosf(0.5f * subtriangle_solid_angle), sinf(0.5f * subtriangle_solid_angle));
    vec3 offset = vertices[0] * (parameters[0] * cos_sin.x - parameters[1] * cos_sin.y) + vertices[2] * (parameters[2] * cos_sin.y);
    float k = 2.0f * (dot(vertices[0], offset) / dot(offset, offset));
    vec3 new_vertex_2(fmaf(k, offset.x, -vertices[0].x), fmaf(k, offset.y, -vertices[0].y), fmaf(k, offset.z, -vertices[0].z));
    float s2 = dot(vertices[1], new_vertex_2);
    float s = mix_fma(1.0f, s2, random_numbers[1]);
    float denominator = fmaf(-s2, s2, 1.0f);
    float t_normed = sqrtf(fmaf(-s, s, 1.0f) / denominator);
    t_normed = (denominator > 0.0f) ? t_normed : random_numbers[1];
    return fmaf(-t_normed, s2, s) * vertices[1] + t_normed * new_vertex_2;
}

// ---------------------------------------------------------------- sun
// rendering/lights/sun.glsl:9-15
static inline vec3 sample_sun_dir(vec3 sun_dir, float cos_radius, vec2 sampl) {
    float phi = 2.0f * M_PIf * sampl.x;
    float cosTheta = mix(1.0f, cos_radius, sampl.y);
    float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    return ortho_frame(sun_dir) * vec3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta);
}
// sun.glsl:17-20
static inline float sample_sun_dir_pdf(float cos_radius) {
    float solid_angle = 2.0f * M_PIf * (1.0f - cos_radius);
    return 1.0f / solid_angle;
}

// ---------------------------------------------------------------- sky
// rendering/lights/sky_model_arhosek/sky_model.glsl:40-59
static inline vec3 skymodel_radiance(const RptrSkyModelParams &state, const vec3 sun_dir, const vec3 view_dir) {
    float cosTheta = clampf(view_dir.y, 0.0f, 1.0f);
    float cosGamma = clampf(dot(view_dir, sun_dir), -1.0f, 1.0f);
    float gamma = acosf(cosTheta);
    auto cfg = [&](int i) { return vec3(state.configs[i][0], state.configs[i][1], state.configs[i][2]); };
    const vec3 c4g = cfg(4) * gamma;
    const vec3 expM(expf(c4g.x), expf(c4g.y), expf(c4g.z));
    const float rayM = cosGamma * cosGamma;
    const vec3 c8 = cfg(8);
    const vec3 mie_base = vec3(1.0f) + c8 * c8 - 2.0f * c8 * cosGamma;
    // pow(x, 1.5) evaluated as x*sqrt(x) on host and device alike
    const vec3 mie_den(mie_base.x * sqrtf(mie_base.x), mie_base.y * sqrtf(mie_base.y), mie_base.z * sqrtf(mie_base.z));
    const vec3 mieM = vec3(1.0f + cosGamma * cosGamma) / mie_den;
    const float zenith = sqrtf(cosTheta);
    const vec3 c1d = cfg(1) / (cosTheta + 0.01f);
    const vec3 e1(expf(c1d.x), expf(c1d.y), expf(c1d.z));
    const vec3 radiance_coeffs =
        (vec3(1.0f) + cfg(0) * e1) * ((((cfg(2) + cfg(3) * expM) + cfg(5) * rayM) + cfg(6) * mieM) + cfg(7) * zenith);
    return radiance_coeffs * vec3(state.radiances[0], state.radiances[1], state.radiances[2]) * 0.01f;
}

// rendering/mc/nee_interface.glsl:11-15
static inline float nee_mis_heuristic(float n_f, float pdf_f, float n_g, float pdf_g) {
    float f = n_f * pdf_f;
    float g = n_g * pdf_g;
    return f / (f + g);
}

// ---------------------------------------------------------------- footprint (USE_MIPMAPPING helpers)
// rendering/rt/footprint.glsl:10-15
static inline mat2 dpdxy_to_footprint(vec3 ray_dir, vec3 dpdx, vec3 dpdy) {
    vec3 t, b;
    ortho_basis(t, b, ray_dir);
    // F = transpose(mat2x3(t,b)) * mat2x3(dpdx,dpdy)
    mat2 F(vec2(dot(t, dpdx), dot(b, dpdx)), vec2(dot(t, dpdy), dot(b, dpdy)));
    return F * transpose(F);
}

// rendering/rt/footprint.glsl:28-43
static inline mat2 transform_footprint(vec3 dst_ray_dir, const mat3 &T, vec3 src_ray_dir, const mat2 &F) {
    vec3 t, b;
    ortho_basis(t, b, src_ray_dir);
    const vec3 Tt = T * t, Tb = T * b; // mat2x3 T2 = T * mat2x3(t, b)
    ortho_basis(t, b, dst_ray_dir);
    const mat2 T3(vec2(dot(t, Tt), dot(b, Tt)), vec2(dot(t, Tb), dot(b, Tb))); // transpose(mat2x3(t, b)) * T2
    return T3 * F * transpose(T3);
}
static inline mat2 reflect_footprint(vec3 dst_ray_dir, vec3 src_ray_dir, const mat2 &F) {
    const vec3 n = normalize(dst_ray_dir - src_ray_dir);
    // mat3(1) - 2 outerProduct(n, n), column by column
    const mat3 R(vec3(1.0f - 2.0f * (n.x * n.x), 0.0f - 2.0f * (n.y * n.x), 0.0f - 2.0f * (n.z * n.x)),
                 vec3(0.0f - 2.0f * (n.x * n.y), 1.0f - 2.0f * (n.y * n.y), 0.0f - 2.0f * (n.z * n.y)),
                 vec3(0.0f - 2.0f * (n.x * n.z), 0.0f - 2.0f * (n.y * n.z), 1.0f - 2.0f * (n.z * n.z)));
    return transform_footprint(dst_ray_dir, R, src_ray_dir, F);
}
// rendering/rt/footprint.glsl:45-63
static inline void footprint_to_dpdxy(vec3 &dpdx, vec3 &dpdy, vec3 ray_dir, const mat2 &F) {
    const float B = F[0][0] + F[1][1];
    const float C = F[0][0] * F[1][1] - F[0][1] * F[1][0];
    const float D = sqrtf(B * B * 0.25f - C);
    const vec2 ev(0.5f * B - D, 0.5f * B + D);
    mat2 X;
    if (fabsf(F[0][1]) > 3.0e-39f) {
        X[0] = vec2(F[1][0], ev.x - F[0][0]);
        X[1] = vec2(ev.y - F[1][1], F[0][1]);
    } else
        X = mat2(1.0f);
    vec3 t, b;
    ortho_basis(t, b, ray_dir);
    const vec2 x0 = normalize(X[0]) * sqrtf(ev.x), x1 = normalize(X[1]) * sqrtf(ev.y);
    dpdx = t * x0.x + b * x0.y; // mat2x3(t, b) * v
    dpdy = t * x1.x + b * x1.y;
}

} // namespace orc
