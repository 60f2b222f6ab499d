// lights.hpp -- host-side emitter preparation for next-event estimation: the scene-load step the reference runs once per scene on
// the CPU, in the reference's own language. The C++ twin of lights.py; tests compare the two.
//   collect_emitters              librender/lights.cpp:14-73
//   estimate_normalized_radiance  librender/lights.cpp:166-199 (triangle_solid_angle :127-163)
//   trim_dim_emitters             librender/lights.cpp:201-217
//   equalize_emitter_bins         librender/lights.cpp:220-349
//   update_light_sampling         librender/lights.cpp:75-90
//   halton2                       util/compute_util.h:19-33
// In a drop-in build librender provides these; this header lets the C++ hosts prepare a scene read from a .vks
// file without the reference and without Python. Plain float arithmetic in the reference's order (build without -ffast-math).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

#include "scene_dump.hpp"

namespace rptr {
namespace lights {

inline float halton2(uint32_t index) { // base-2 radical inverse through bit reversal, mantissa trick
    index = (index << 16) | (index >> 16);
    index = ((index & 0x00FF00FFu) << 8) | ((index & 0xFF00FF00u) >> 8);
    index = ((index & 0x0F0F0F0Fu) << 4) | ((index & 0xF0F0F0F0u) >> 4);
    index = ((index & 0x33333333u) << 2) | ((index & 0xCCCCCCCCu) >> 2);
    index = ((index & 0x55555555u) << 1) | ((index & 0xAAAAAAAAu) >> 1);
    const uint32_t u = 0x3F800000u | (index >> 9);
    float f;
    std::memcpy(&f, &u, 4);
    return f - 1.0f;
}

struct V3 {
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
inline V3 normalize(V3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float luminance(const float c[3]) { return (0.2126f * c[0] + 0.7152f * c[1]) + 0.0722f * c[2]; }

// lights.cpp:127-163: solid angle of the spherical triangle of three unit vectors (Householder reflection to the x axis, then a 2x2
// determinant), with a true atan on the host
inline float triangle_solid_angle(V3 v0, V3 v1, V3 v2) {
    const float householder_sign = v0.x > 0.0f ? -1.0f : 1.0f;
    const float s = 1.0f / (std::fabs(v0.x) + 1.0f);
    const float hy = v0.y * s, hz = v0.z * s;
    const float dot_0_1 = dot(v0, v1), dot_0_2 = dot(v1, v2), dot_1_2 = dot(v0, v2);
    const float dh0 = std::fmaf(-householder_sign, v1.x, dot_0_1), dh2 = std::fmaf(-householder_sign, v2.x, dot_1_2);
    const float m00 = std::fmaf(-dh0, hy, v1.y), m01 = std::fmaf(-dh0, hz, v1.z);
    const float m10 = std::fmaf(-dh2, hy, v2.y), m11 = std::fmaf(-dh2, hz, v2.z);
    const float simplex_volume = std::fabs(m00 * m11 - m10 * m01);
    const float tangent = simplex_volume / ((1.0f + dot_0_1) + (dot_0_2 + dot_1_2));
    const float offset = tangent < 0.0f ? 3.14159265358979323846f : 0.0f;
    return 2.0f * (std::atan(tangent) + offset);
}

inline V3 dequantize_position(uint64_t w, const float sc[3], const float of[3]) { // librender/dequantize.glsl:8-21
    return {float(uint32_t(w) & 0x1FFFFFu) * sc[0] + of[0], float(uint32_t(w >> 21) & 0x1FFFFFu) * sc[1] + of[1],
            float(uint32_t(w >> 42) & 0x1FFFFFu) * sc[2] + of[2]};
}

// lights.cpp:14-73: every triangle of every instance whose material emits, in world space; the emitters of an instance go IN FRONT of
// those collected so far (emitters.insert(emitters.begin(), ...)); a parameterized mesh found without emitters is not looked at again
inline std::vector<RptrTriLightData> collect_emitters(const SceneDump &s) {
    std::vector<RptrTriLightData> emitters;
    std::set<uint32_t> nonemissive;
    for (const RptrInstanceDesc &inst : s.instances) {
        const uint32_t pm_id = inst.parameterized_mesh;
        if (nonemissive.count(pm_id)) continue;
        const RptrParameterizedMeshDesc &pm = s.pmeshes[pm_id];
        const RptrMeshDesc &mesh = s.meshes[pm.mesh];
        std::vector<RptrTriLightData> next;
        size_t tri_base = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const RptrGeometryDesc &g = s.geometries[mesh.first_geometry + j];
            const int32_t offs = pm.material_offsets[j];
            const bool per_tri = pm.tri_material_ids != nullptr;
            if (!per_tri && !(s.materials[(size_t)offs].emission_intensity > 0.0f)) {
                tri_base += g.num_tris;
                continue;
            }
            for (uint32_t t = 0; t < g.num_tris; ++t) {
                const RptrBaseMaterial &mat = s.materials[(size_t)(offs + (per_tri ? (int32_t)pm.tri_material_ids[tri_base + t] : 0))];
                if (!(mat.emission_intensity > 0.0f)) continue;
                RptrTriLightData e;
                for (int c = 0; c < 3; ++c) e.radiance[c] = mat.emission_intensity * mat.base_color[c];
                const float *M = inst.transform; // 3x4 row-major object -> world; association of glm's mat4 * vec4: (m0 x + m1 y) + (m2 z + m3)
                float *dst[3] = {e.v0, e.v1, e.v2};
                for (int k = 0; k < 3; ++k) {
                    const V3 p = dequantize_position(g.qpos[3 * (size_t)t + k], g.quantized_scaling, g.quantized_offset);
                    for (int r = 0; r < 3; ++r) dst[k][r] = (M[4 * r + 0] * p.x + M[4 * r + 1] * p.y) + (M[4 * r + 2] * p.z + M[4 * r + 3]);
                }
                next.push_back(e);
            }
            tri_base += g.num_tris;
        }
        if (!next.empty())
            emitters.insert(emitters.begin(), next.begin(), next.end());
        else
            nonemissive.insert(pm_id);
    }
    return emitters;
}

// lights.cpp:166-199: luminance x solid angle seen from `min_perceived_receiver_dist` above the centre. (The reference divides by
// M_2_PI = 2 / pi; reproduced.)
inline std::vector<float> estimate_normalized_radiance(const std::vector<RptrTriLightData> &emitters, float min_perceived_receiver_dist) {
    std::vector<float> out(emitters.size(), 0.0f);
    for (size_t i = 0; i < emitters.size(); ++i) {
        const RptrTriLightData &e = emitters[i];
        const V3 v0{e.v0[0], e.v0[1], e.v0[2]}, v1{e.v1[0], e.v1[1], e.v1[2]}, v2{e.v2[0], e.v2[1], e.v2[2]};
        const V3 n = normalize(cross(v1 - v0, v2 - v0));
        const float ln = std::sqrt(dot(n, n));
        if (!(std::fabs(ln - 1.0f) < 0.05f)) continue; // degenerate triangle
        const V3 sum = (v0 + v1) + v2;
        const V3 cen{sum.x / 3.0f, sum.y / 3.0f, sum.z / 3.0f};
        const V3 o = n * min_perceived_receiver_dist;
        const float sa = triangle_solid_angle(normalize((v0 - cen) - o), normalize((v1 - cen) - o), normalize((v2 - cen) - o));
        // `luminance(..) * (solid_angle / M_2_PI)`: M_2_PI is a double, so quotient AND product are doubles, rounded once on the store (lights.cpp:195)
        out[i] = float(double(luminance(e.radiance)) * (double(sa) / 0.63661977236758134308));
    }
    return out;
}

// lights.cpp:220-349: bright emitters are split into clones, the list is shuffled by a Halton sequence and padded with clones drawn
// by importance until every bin of `bin_size` lights carries about the same power
inline void equalize_emitter_bins(std::vector<RptrTriLightData> &emitters, std::vector<float> &radiances, int bin_size) {
    const size_t n = radiances.size();
    if (bin_size <= 1 || n == 0) return;
    struct Bin {
        float radiance;
        long source;
        long split;
    };
    const size_t original_bin_count = (n + size_t(bin_size - 1)) / size_t(bin_size);
    float average_weight = 0.0f;
    for (float r : radiances) average_weight += r;
    average_weight /= float(n);
    std::vector<Bin> bins;
    for (size_t i = 0; i < n; ++i) {
        const float q = radiances[i] / average_weight;
        const double qq = q == q ? std::min(double(q), double(original_bin_count)) : double(original_bin_count);
        const long clones = std::max<long>(qq >= 0.0 ? long(uint32_t(qq)) : 0, 1);
        for (long c = 0; c < clones; ++c) bins.push_back({radiances[i] / float(clones), (long)i, clones});
    }
    auto reshuffle = [](std::vector<Bin> &b) {
        const size_t count = b.size();
        std::vector<Bin> out(count);
        for (size_t index = 0; index < count; ++index) {
            size_t src = size_t(uint32_t(halton2((uint32_t)index) * float(count)));
            for (;;) {
                if (src >= count) src = 0;
                if (b[src].source == -1)
                    ++src;
                else
                    break;
            }
            out[index] = b[src];
            b[src].source = -1;
        }
        b.swap(out);
    };
    auto measure_equality = [&](const std::vector<Bin> &b) {
        float mn = 2.0e32f, mx = 0.0f;
        for (size_t i = 0; i < b.size();) {
            float tot = 0.0f;
            for (int j = 0; j < bin_size && i < b.size(); ++j, ++i) tot += b[i].radiance;
            mn = std::min(tot, mn);
            mx = std::max(tot, mx);
        }
        return std::min(mn / mx, 1.0f);
    };
    reshuffle(bins);
    float equality = measure_equality(bins);
    for (int retries = 0; equality < 0.6f && retries < 2; ++retries) {
        std::vector<Bin> postfix(bins.size());
        for (size_t k = 0; k < bins.size(); ++k)
            postfix[k] = k == 0 ? bins[0] : Bin{postfix[k - 1].radiance + bins[k].radiance, bins[k].source, 1};
        postfix[0].split = 1;
        const float total = postfix.back().radiance;
        for (Bin &b : postfix) b.radiance /= total;
        const size_t prev_elements = bins.size();
        const size_t prev_bin_count = (prev_elements + size_t(bin_size - 1)) / size_t(bin_size);
        const size_t padded = (prev_bin_count + 1) * size_t(bin_size);
        uint32_t h = 0;
        while (bins.size() < padded) {
            const float u = halton2(h++);
            // upper_bound: the first element whose cumulative share exceeds u
            size_t it = size_t(std::upper_bound(postfix.begin(), postfix.end(), u, [](float v, const Bin &b) { return double(v) < double(b.radiance); }) - postfix.begin());
            if (it == postfix.size()) it = postfix.size() - 1;
            postfix[it].split += 1;
            bins.push_back({postfix[it].radiance, (long)it, 0});
        }
        for (size_t i = prev_elements; i < padded; ++i) {
            Bin &clone = bins[i];
            Bin &original = bins[(size_t)clone.source];
            const long cc = postfix[(size_t)clone.source].split;
            if (cc > 1) {
                original.radiance /= float(cc);
                original.split *= cc;
                postfix[(size_t)clone.source].split = 1;
            }
            clone = original;
        }
        reshuffle(bins);
        equality = measure_equality(bins);
    }
    std::vector<RptrTriLightData> new_em(bins.size());
    std::vector<float> new_rad(bins.size());
    for (size_t i = 0; i < bins.size(); ++i) {
        new_em[i] = emitters[(size_t)bins[i].source];
        for (int c = 0; c < 3; ++c) new_em[i].radiance[c] /= float(bins[i].split);
        new_rad[i] = bins[i].radiance;
    }
    emitters.swap(new_em);
    radiances.swap(new_rad);
}

// lights.cpp:75-90 from an invalidated BinnedLightSampling
inline std::vector<RptrTriLightData> update_light_sampling(std::vector<RptrTriLightData> emitters, const RptrLightSamplingConfig &cfg) {
    if (emitters.empty()) return emitters;
    std::vector<float> radiances = estimate_normalized_radiance(emitters, cfg.min_perceived_receiver_dist);
    if (cfg.min_radiance > 0.0f) { // trim_dim_emitters, :201-217
        std::vector<RptrTriLightData> e2;
        std::vector<float> r2;
        for (size_t i = 0; i < emitters.size(); ++i)
            if (radiances[i] >= cfg.min_radiance) {
                e2.push_back(emitters[i]);
                r2.push_back(radiances[i]);
            }
        emitters.swap(e2);
        radiances.swap(r2);
    }
    equalize_emitter_bins(emitters, radiances, cfg.bin_size);
    return emitters;
}

// RenderBinnedLightsVulkan::update_scene_from_backend (render_binned_lights.cpp:68-87): fills s.lights
inline void prepare_lights(SceneDump &s) { s.lights = update_light_sampling(collect_emitters(s), s.lighting); }

} // namespace lights
} // namespace rptr
