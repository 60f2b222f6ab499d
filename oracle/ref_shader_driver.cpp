// ref_shader_driver.cpp -- TEST INFRASTRUCTURE, built only where a real GLM is present (`make -C oracle ref_shaders GLM_ROOT=<dir holding glm/glm.hpp>`).
//
// The reference's shader library is dual-language: the .glsl files under rendering/ compile as C++ against GLM
// (rendering/tests/compile.cpp:1-41 is the reference's own proof: it includes them in exactly this way). This driver includes the
// reference's files WHERE THEY LIE (no copies, no stand-in headers: without GLM it does not build, and the recipe says so), calls the
// functions that decide a pixel on seeded inputs and prints inputs + outputs as JSON: the golden vectors that pin the oracle
// (oracle/oshade.h) and, through it, the device code (csrc/dshade.h). SURVEY.md section 8(c) lists twelve vector groups; this driver makes
// ten of them and the NEE half of the eleventh (the whole eleventh -- one shading step of the megakernel -- is ref_shade_driver.cpp, a separate program because it
// needs the megakernel's macro state); the other two are pinned elsewhere against the reference's own compiled code:
//    (1) "rng"            get_lcg_rng / lcg_randomf (rendering/pointsets/lcg_rng.glsl): 64 (index, frame offset, pixel) tuples -> state + 8 draws,
//                         in the order the draws are made (section 7.2-2: every vector below lists its random numbers left to right)
//    (2) "dequant"        DEQUANTIZE_POSITION / dequantize_normal / dequantize_uv (librender/dequantize.glsl) on 256 words
//    (3) "hit_attributes" calc_hit_attributes, the quantised overload the megakernel uses (rendering/rt/hit.glsl:58-128,162-203): 256 triangles x
//                         barycentrics, with / without normals and uvs, identity and random normals_to_world
//    (4) "footprint"      dpdxy_to_footprint, footprint_to_dpdxy, reflect_footprint (rendering/rt/footprint.glsl)
//    (5) "gltf", "simple" sample_gltf_brdf / gltf_bsdf / gltf_wpdf (bsdfs/gltf_bsdf.glsl:294-650); sample_simple_brdf / simple_bsdf / simple_pdf
//    (6) "tri_lights"     sample_tri_lights over a fixed light table + approx_tri_lights_pdf (mc/lights_linear.glsl:19-137)
//    (7) "sun"            sample_sun_dir, sample_sun_dir_pdf (lights/sun.glsl)
//    (8) "sky"            skymodel_radiance on a direction grid for the printed SkyModelParams (lights/sky_model_arhosek/sky_model.glsl); the FIT
//                         that makes such parameters is pinned by oracle/ref_sky_driver.cpp (sky_model.cpp compiled unmodified: tests/test_sky_fit.py)
//   (10) "srgb"           linear_to_srgb (util.glsl:25-28) on 256 values (the running mean of vulkan/accumulate.glsl is GLSL-only: images pin it)
//   (11) equalize_emitter_bins: ref_lights_driver.cpp (librender/lights.cpp compiled unmodified, update_light_sampling on seven emitter sets); its Halton
//                         table is pinned by tests/test_oracle.py against librender/halton.h compiled
//   (12) vkr_quantize_transform / dequantize: tests/test_vks.py against ext/libvkr/src/vkr.c compiled unmodified (oracle/_ref/libvkr_ref.so)
//    (9) "nee"           of shade_base_material's end-to-end chain the part that needs no texture unit: sample_direct_light (mc/nee.glsl:32-90 -- sun or
//                         triangle lights, MIS, strict normals; visibility stubbed "visible" as compile.cpp:39 does). unpack_material + the BSDF
//                         sample + the termination tests around it need the megakernel's sampler / SCENE_GET_* / DIM_* environment: ref_shade_driver.cpp.
// Output: tests/golden/ref_shaders.json (written by `make ref_shaders`); read by tests/test_ref_shaders.py on the CPU (oracle) and, -m gpu,
// through images. Until that file exists those tests skip with this reason.
//
// NOTE for whoever runs this first: it has never been compiled against a real GLM (the build container has none). It follows compile.cpp's
// include order and macro hooks to the letter; if the reference's headers need one more hook, add it HERE, not in a stand-in header.
#include <glm/glm.hpp>

#include <cstdint>
#include <cstdio>
#include <random>

namespace ref_shaders {

using namespace glm;
#include "rendering/language.hpp"

#include "rendering/pointsets/lcg_rng.glsl"

#include "rendering/util.glsl"

#include "rendering/bsdfs/base_material.h.glsl"

#define NO_MATERIAL_REGISTRATION
namespace gltf {
#include "rendering/bsdfs/gltf_bsdf.glsl"
}

#include "rendering/lights/tri.glsl"
#include "rendering/lights/sun.glsl"
#include "rendering/lights/sky_model_arhosek/sky_model.glsl"

#define DEFAULT_GEOMETRY_BUFFER_TYPES
#include "rendering/rt/hit.glsl" // (+ rt/geometry.h.glsl, librender/dequantize.glsl)
#include "rendering/rt/footprint.glsl"

namespace simple {
#include "rendering/bsdfs/simple_bsdf.glsl"
}

static const int global_num_lights = 40; // (not `num_lights`: sample_tri_lights has a local of that name which these macros initialise, compile.cpp:21)
static TriLightData lights[global_num_lights + 16] = {}; // (padded with a zeroed bin: sample_tri_lights may read light_id == bin_end)
static int bin_size = 16;

#define SCENE_GET_LIGHT_SOURCE(light_id) decode_tri_light(lights[light_id])
#define SCENE_GET_LIGHT_SOURCE_COUNT() int(global_num_lights)
// the megakernel's definitions (vulkan/pt_megakernel.glsl:101-103) over this driver's table
#define BINNED_LIGHTS_BIN_MAX_SIZE 16
#define BINNED_LIGHTS_BIN_SIZE int(bin_size)
#define SCENE_GET_BINNED_LIGHTS_BIN_COUNT() ((int(global_num_lights) + (bin_size - 1)) / int(bin_size))

// next-event estimation end to end (rendering/mc/nee.glsl:32-90 sample_direct_light: sun / triangle-light choice, MIS weight, strict normals,
// the visibility query -- answered "visible" here, as rendering/tests/compile.cpp:39 does) over the glTF material; nee.glsl pulls in
// lights_sun.glsl, lights_linear.glsl (the binned-RIS sampler, with the macros above) and nee_interface.glsl
namespace binned {
struct SceneParamsOfTheDriver { // the members nee.glsl / nee_interface.glsl read of the megakernel's scene_params (vulkan/gpu_params.glsl:120-131)
    glm::vec4 sun_radiance;     // .w: the probability of sampling the sun
    glm::vec3 sun_dir;
    float sun_cos_angle;
};
static SceneParamsOfTheDriver scene_params;
#define MATERIAL_TYPE gltf::GLTFMaterial
#define eval_bsdf(mat, hit, w_o, w_i) gltf::gltf_bsdf(mat, hit.n, w_o, w_i, hit.v_x, hit.v_y)
#define eval_bsdf_wpdf(mat, hit, w_o, w_i) gltf::gltf_wpdf(mat, hit.n, w_o, w_i, hit.v_x, hit.v_y)
#include "rendering/mc/nee.glsl"
inline bool raytrace_test_visibility(const vec3 from, const vec3 dir, float dist) { return true; }
}

} // namespace ref_shaders

static void p3(const char *k, const glm::vec3 &v, bool last = false) { std::printf("\"%s\": [%.9g, %.9g, %.9g]%s", k, v.x, v.y, v.z, last ? "" : ", "); }

int main() {
    using namespace ref_shaders;
    std::mt19937 gen(20240917u);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    auto unit = [&]() { // uniform on the sphere
        const float z = 2.0f * U(gen) - 1.0f, phi = 6.2831853f * U(gen), r = std::sqrt(std::max(0.0f, 1.0f - z * z));
        return glm::vec3(r * std::cos(phi), r * std::sin(phi), z);
    };
    std::printf("{\n\"generator\": \"oracle/ref_shader_driver.cpp over the reference's rendering/*.glsl compiled against GLM\",\n");
    // ---- (1) RNG. "rng": SURVEY 8a2's known answer (index 3, frame offset 7, pixel (10, 20) of a 256 x 256 frame -> state 1349923967, first draw
    // 0.314303666: tests/test_oracle.py holds the oracle to it); "rng_table": 64 tuples, the state after seeding and the first eight draws in order
    {
        LCGRand rng = get_lcg_rng(3u, 7u, glm::uvec4(10u, 20u, 256u, 256u));
        const uint32_t s0 = rng.state;
        const float f0 = lcg_randomf(rng);
        std::printf("\"rng\": {\"state\": %u, \"first\": %.9g},\n\"rng_table\": [\n", s0, f0);
        for (int i = 0; i < 64; ++i) {
            const uint32_t index = gen() % 4096u, frame = gen() % 100000u, px = gen() % 1920u, py = gen() % 1080u;
            LCGRand r = get_lcg_rng(index, frame, glm::uvec4(px, py, 1920u, 1080u));
            std::printf("{\"index\": %u, \"frame_offset\": %u, \"pixel\": [%u, %u], \"dims\": [1920, 1080], \"state\": %u, \"draws\": [", index, frame, px, py, r.state);
            for (int k = 0; k < 8; ++k) std::printf("%.9g%s", lcg_randomf(r), k < 7 ? ", " : "");
            std::printf("]}%s\n", i < 63 ? "," : "");
        }
        std::printf("],\n");
    }
    // ---- (2) dequantisation: 256 random words (21 : 21 : 21 bits of position; oct normal | uv)
    {
        const glm::vec3 scaling(1.0f / 1024.0f, 3.0f / 2097151.0f, 0.25f), offset(-7.5f, 0.125f, 100.0f);
        std::printf("\"dequant\": {\"scaling\": [%.9g, %.9g, %.9g], \"offset\": [%.9g, %.9g, %.9g], \"words\": [\n", scaling.x, scaling.y, scaling.z, offset.x, offset.y, offset.z);
        for (int i = 0; i < 256; ++i) {
            const uint64_t w = (uint64_t(gen()) << 32) | uint64_t(gen());
            const glm::vec3 p = DEQUANTIZE_POSITION(w, scaling, offset);
            const glm::vec3 n = dequantize_normal(uint32_t(w));
            const glm::vec2 uv = dequantize_uv(uint32_t(w >> 32));
            std::printf("{\"lo\": %u, \"hi\": %u, ", uint32_t(w), uint32_t(w >> 32));
            p3("position", p); p3("normal", n);
            std::printf("\"uv\": [%.9g, %.9g]}%s\n", uv.x, uv.y, i < 255 ? "," : "");
        }
        std::printf("]},\n");
    }
    // ---- (3) hit attributes: the quantised overload (hit.glsl:162-203) over a three-vertex stream
    {
        std::printf("\"hit_attributes\": [\n");
        const int n_hit = 256;
        for (int i = 0; i < n_hit; ++i) {
            const glm::vec3 va(4.0f * U(gen) - 2.0f, 4.0f * U(gen) - 2.0f, 4.0f * U(gen) - 2.0f);
            const glm::vec3 vb = va + (i % 9 == 0 ? 1e-3f : 1.0f) * unit(), vc = va + (i % 11 == 0 ? 1e-3f : 1.0f) * unit();
            uint64_t nuv[3];
            for (int k = 0; k < 3; ++k) nuv[k] = (uint64_t(gen()) << 32) | uint64_t(gen());
            const bool has_normals = (i & 1) != 0, has_uvs = (i & 2) != 0;
            glm::mat3 n2w(1.0f);
            if (i & 4) { // a rotation x non-uniform scale, as transpose(mat3(world_to_object)) of an instance would be
                const glm::vec3 ax = unit();
                glm::vec3 ay = normalize(cross(ax, unit())), az = cross(ax, ay);
                n2w = glm::mat3(ax * (0.5f + U(gen)), ay * (0.5f + U(gen)), az * (0.5f + U(gen)));
            }
            float bu = U(gen), bv = U(gen);
            if (bu + bv > 1.0f) { bu = 1.0f - bu; bv = 1.0f - bv; }
            const float t = 0.1f + 50.0f * U(gen);
            QuantizedNormalUVBuffer nbuf{nuv};
            uint32_t ids4 = 0x03020100u;
            MaterialIDBuffer mbuf{&ids4};
            const int material_in = (i % 5 == 0) ? -3 : 7; // (< 0: per-triangle ids, offset -material_in - 1, hit.glsl:49-56)
            const RTHit h = calc_hit_attributes(t, 0u, glm::vec2(bu, bv), glm::mat3(va, vb, vc), glm::uvec3(0u, 1u, 2u), n2w, nbuf, has_normals, has_uvs, material_in, mbuf);
            std::printf("{");
            p3("va", va); p3("vb", vb); p3("vc", vc);
            std::printf("\"nuv\": [[%u, %u], [%u, %u], [%u, %u]], \"has_normals\": %d, \"has_uvs\": %d, ", uint32_t(nuv[0]), uint32_t(nuv[0] >> 32), uint32_t(nuv[1]),
                        uint32_t(nuv[1] >> 32), uint32_t(nuv[2]), uint32_t(nuv[2] >> 32), int(has_normals), int(has_uvs));
            std::printf("\"normals_to_world\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g], ", n2w[0][0], n2w[0][1], n2w[0][2], n2w[1][0], n2w[1][1], n2w[1][2],
                        n2w[2][0], n2w[2][1], n2w[2][2]); // column by column
            std::printf("\"t\": %.9g, \"bary\": [%.9g, %.9g], \"material_in\": %d, ", t, bu, bv, material_in);
            p3("normal", h.normal); p3("geo_normal", h.geo_normal); p3("tangent", h.tangent);
            std::printf("\"dist\": %.9g, \"material_id\": %d, \"bitangent_l\": %.9g, \"uv\": [%.9g, %.9g]}%s\n", h.dist, h.material_id, h.bitangent_l, h.uv.x, h.uv.y,
                        i + 1 < n_hit ? "," : "");
        }
        std::printf("],\n");
    }
    // ---- (4) texture footprints
    {
        std::printf("\"footprint\": [\n");
        const int n_fp = 128;
        for (int i = 0; i < n_fp; ++i) {
            const glm::vec3 dir = unit(), dst = unit();
            const glm::vec3 dpdx = 0.01f * (0.2f + U(gen)) * unit(), dpdy = 0.01f * (0.2f + U(gen)) * unit();
            const glm::mat2 F = dpdxy_to_footprint(dir, dpdx, dpdy);
            glm::vec3 bx(0.0f), by(0.0f);
            footprint_to_dpdxy(bx, by, dir, F);
            const glm::mat2 R = reflect_footprint(dst, dir, F);
            std::printf("{");
            p3("dir", dir); p3("dst_dir", dst); p3("dpdx", dpdx); p3("dpdy", dpdy);
            std::printf("\"footprint\": [%.9g, %.9g, %.9g, %.9g], ", F[0][0], F[0][1], F[1][0], F[1][1]);
            p3("back_dpdx", bx); p3("back_dpdy", by);
            std::printf("\"reflected\": [%.9g, %.9g, %.9g, %.9g]}%s\n", R[0][0], R[0][1], R[1][0], R[1][1], i + 1 < n_fp ? "," : "");
        }
        std::printf("],\n");
    }
    // ---- (5b) the Lambert BSDF of the diffuse-only configuration (bsdfs/simple_bsdf.glsl:44-94)
    {
        std::printf("\"simple\": [\n");
        const int n_s = 256;
        for (int i = 0; i < n_s; ++i) {
            simple::SimpleMaterial m = {};
            m.base_color = glm::vec3(U(gen), U(gen), U(gen));
            m.roughness = 1.0f;
            m.ior = 1.0f;
            const glm::vec3 n = unit();
            glm::vec3 wo = unit();
            if (dot(n, wo) < 0.0f && (i % 7)) wo = -wo;
            const glm::vec2 u(U(gen), U(gen));
            glm::vec3 wi(0.0f);
            float pdf = 0.0f, mis = 0.0f;
            const glm::vec3 w = simple::sample_simple_brdf(m, n, wo, wi, pdf, mis, u);
            const glm::vec3 wie = unit();
            const glm::vec3 f = simple::simple_bsdf(m, n, wo, wie);
            const float fp = simple::simple_pdf(m, n, wo, wie);
            std::printf("{");
            p3("base_color", m.base_color); p3("n", n); p3("wo", wo);
            std::printf("\"u\": [%.9g, %.9g], ", u.x, u.y);
            p3("wi", wi); p3("weight", w);
            std::printf("\"pdf\": %.9g, \"mis_pdf\": %.9g, ", pdf, mis);
            p3("wi_eval", wie); p3("f", f);
            std::printf("\"wpdf\": %.9g}%s\n", fp, i + 1 < n_s ? "," : "");
        }
        std::printf("],\n");
    }
    // ---- (7) the sun's cone
    {
        std::printf("\"sun\": [\n");
        for (int i = 0; i < 64; ++i) {
            const glm::vec3 sd = unit();
            const float cos_radius = std::cos(0.00465f * (1.0f + 20.0f * U(gen)));
            const glm::vec2 u(U(gen), U(gen));
            const glm::vec3 d = sample_sun_dir(sd, cos_radius, u);
            const float pdf = sample_sun_dir_pdf(sd, cos_radius, d);
            std::printf("{");
            p3("sun_dir", sd);
            std::printf("\"cos_radius\": %.9g, \"u\": [%.9g, %.9g], ", cos_radius, u.x, u.y);
            p3("dir", d);
            std::printf("\"pdf\": %.9g}%s\n", pdf, i < 63 ? "," : "");
        }
        std::printf("],\n");
    }
    // ---- (8) sky radiance on a 16 x 8 grid of directions, for parameters of the shape the host fit produces (their values: any; printed)
    {
        SkyModelParams sky;
        for (int k = 0; k < 9; ++k) sky.configs[k] = glm::vec4(-1.2f + 0.3f * k + 0.1f * U(gen), -0.4f + 0.1f * k + 0.1f * U(gen), 0.2f * k - 0.5f + 0.1f * U(gen), 0.0f);
        sky.configs[8] = glm::vec4(0.35f, 0.4f, 0.45f, 0.0f); // (|g| < 1: the Mie term's base stays positive)
        sky.radiances = glm::vec4(9.0f, 11.0f, 14.0f, 0.0f);
        const glm::vec3 sd = normalize(glm::vec3(0.3f, 0.8f, 0.5f));
        std::printf("\"sky\": {\"configs\": [");
        for (int k = 0; k < 9; ++k) std::printf("[%.9g, %.9g, %.9g, %.9g]%s", sky.configs[k].x, sky.configs[k].y, sky.configs[k].z, sky.configs[k].w, k < 8 ? ", " : "");
        std::printf("], \"radiances\": [%.9g, %.9g, %.9g, %.9g], ", sky.radiances.x, sky.radiances.y, sky.radiances.z, sky.radiances.w);
        p3("sun_dir", sd);
        std::printf("\"grid\": [\n");
        for (int j = 0; j < 8; ++j)
            for (int i = 0; i < 16; ++i) {
                const float theta = 1.5607963f * (float(j) + 0.5f) / 8.0f, phi = 6.2831853f * float(i) / 16.0f;
                const glm::vec3 v(std::sin(theta) * std::cos(phi), std::cos(theta), std::sin(theta) * std::sin(phi));
                const glm::vec3 r = skymodel_radiance(sky, sd, v);
                std::printf("{");
                p3("dir", v); p3("radiance", r, true);
                std::printf("}%s\n", (j == 7 && i == 15) ? "" : ",");
            }
        std::printf("]},\n");
    }
    // ---- (10) the display transfer function
    {
        std::printf("\"srgb\": [");
        for (int i = 0; i < 256; ++i) {
            const float x = i < 16 ? 0.0031308f * float(i) / 8.0f : U(gen) * (i % 3 ? 1.0f : 4.0f);
            std::printf("[%.9g, %.9g]%s", x, linear_to_srgb(x), i < 255 ? ", " : "");
        }
        std::printf("],\n");
    }
    // ---- glTF BSDF
    std::printf("\"gltf\": [\n");
    const int n_bsdf = 512;
    for (int i = 0; i < n_bsdf; ++i) {
        gltf::GLTFMaterial m = {};
        m.base_color = glm::vec3(U(gen), U(gen), U(gen));
        m.metallic = (i % 3 == 0) ? 0.0f : (i % 3 == 1 ? 1.0f : U(gen));
        m.specular = U(gen);
        m.roughness = (i % 5 == 0) ? 0.02f : U(gen);
        m.ior = 1.0f + U(gen);
        m.flags = 0u;
        glm::vec3 n = unit(), wo = unit();
        if (dot(n, wo) < 0.0f && (i % 7)) wo = -wo; // (every seventh: from below the surface)
        glm::vec3 vx, vy;
        ortho_basis(vx, vy, n);
        const glm::vec2 u_dir(U(gen), U(gen)), u_lobe(U(gen), U(gen));
        glm::vec3 wi(0.0f);
        float pdf = 0.0f, mis = 0.0f;
        const glm::vec3 w = gltf::sample_gltf_brdf(m, n, wo, wi, pdf, mis, u_dir, u_lobe, vx, vy);
        const glm::vec3 wi_e = unit();
        const glm::vec3 f = gltf::gltf_bsdf(m, n, wo, wi_e, vx, vy);
        const float fp = gltf::gltf_wpdf(m, n, wo, wi_e, vx, vy);
        std::printf("{");
        p3("base_color", m.base_color);
        std::printf("\"metallic\": %.9g, \"specular\": %.9g, \"roughness\": %.9g, \"ior\": %.9g, ", m.metallic, m.specular, m.roughness, m.ior);
        p3("n", n); p3("wo", wo);
        std::printf("\"u\": [%.9g, %.9g, %.9g, %.9g], ", u_dir.x, u_dir.y, u_lobe.x, u_lobe.y);
        p3("wi", wi); p3("weight", w);
        std::printf("\"pdf\": %.9g, \"mis_pdf\": %.9g, ", pdf, mis);
        p3("wi_eval", wi_e); p3("f", f);
        std::printf("\"wpdf\": %.9g}%s\n", fp, i + 1 < n_bsdf ? "," : "");
    }
    std::printf("],\n");
    // ---- triangle lights: a table of small emitters above the origin, queries below them
    for (int k = 0; k < global_num_lights; ++k) {
        const glm::vec3 c(6.0f * U(gen) - 3.0f, 2.0f + U(gen), 6.0f * U(gen) - 3.0f);
        const glm::vec3 a = c + 0.3f * unit(), b = c + 0.3f * unit(), d = c + 0.3f * unit();
        TriLightData &t = lights[k];
        t.v0_x = a.x; t.v0_y = a.y; t.v0_z = a.z; t.v1_x = b.x; t.v1_y = b.y; t.v1_z = b.z; t.v2_x = d.x; t.v2_y = d.y; t.v2_z = d.z;
        t.radiance_x = 1.0f + 9.0f * U(gen); t.radiance_y = 1.0f + 9.0f * U(gen); t.radiance_z = 1.0f + 9.0f * U(gen);
    }
    std::printf("\"lights\": [");
    for (int k = 0; k < global_num_lights; ++k)
        std::printf("[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]%s", lights[k].v0_x, lights[k].v0_y, lights[k].v0_z, lights[k].v1_x,
                    lights[k].v1_y, lights[k].v1_z, lights[k].v2_x, lights[k].v2_y, lights[k].v2_z, lights[k].radiance_x, lights[k].radiance_y, lights[k].radiance_z,
                    k + 1 < global_num_lights ? ", " : "");
    std::printf("],\n\"tri_lights\": [\n");
    const int n_lights_q = 256;
    for (int i = 0; i < n_lights_q; ++i) {
        bin_size = (i % 4 == 3) ? 7 : 16;
        const glm::vec3 hp(4.0f * U(gen) - 2.0f, 0.5f * U(gen), 4.0f * U(gen) - 2.0f);
        glm::vec3 hn = unit();
        if (hn.y < 0.0f) hn = -hn;
        const glm::vec2 u_dir(U(gen), U(gen)), u_sel(U(gen), U(gen));
        glm::vec3 ld(0.0f);
        float dist = 0.0f, pdf = 0.0f, mis = 0.0f;
        const glm::vec3 L = binned::sample_tri_lights(hp, hn, u_dir, u_sel, ld, dist, pdf, mis);
        std::printf("{\"bin_size\": %d, ", bin_size);
        p3("p", hp); p3("n", hn);
        std::printf("\"u\": [%.9g, %.9g, %.9g, %.9g], ", u_dir.x, u_dir.y, u_sel.x, u_sel.y);
        p3("radiance_over_pdf", L); p3("dir", ld);
        std::printf("\"dist\": %.9g, \"pdf\": %.9g, \"mis_wpdf\": %.9g}%s\n", dist, pdf, mis, i + 1 < n_lights_q ? "," : "");
    }
    // ---- (9, the part that needs no texture unit) next-event estimation end to end: which light, MIS, strict normals; every shadow ray "visible"
    std::printf("],\n\"nee\": {");
    binned::scene_params.sun_dir = normalize(glm::vec3(0.3f, 0.8f, 0.5f));
    binned::scene_params.sun_cos_angle = std::cos(0.00465f * 4.0f);
    binned::scene_params.sun_radiance = glm::vec4(31000.0f, 29000.0f, 25000.0f, 0.5f);
    p3("sun_dir", binned::scene_params.sun_dir);
    std::printf("\"sun_cos_angle\": %.9g, \"sun_radiance\": [%.9g, %.9g, %.9g, %.9g], \"samples\": [\n", binned::scene_params.sun_cos_angle, binned::scene_params.sun_radiance.x,
                binned::scene_params.sun_radiance.y, binned::scene_params.sun_radiance.z, binned::scene_params.sun_radiance.w);
    const int n_nee = 256;
    bin_size = 16;
    for (int i = 0; i < n_nee; ++i) {
        gltf::GLTFMaterial m = {};
        m.base_color = glm::vec3(U(gen), U(gen), U(gen));
        m.metallic = (i % 3 == 0) ? 0.0f : (i % 3 == 1 ? 1.0f : U(gen));
        m.specular = U(gen);
        m.roughness = 0.1f + 0.9f * U(gen);
        m.ior = 1.0f + U(gen);
        m.flags = 0u;
        binned::InteractionPoint hit; // (bsdfs/hit_point.glsl is first included inside that namespace)
        hit.p = glm::vec3(4.0f * U(gen) - 2.0f, 0.5f * U(gen), 4.0f * U(gen) - 2.0f);
        hit.n = unit();
        if (hit.n.y < 0.0f) hit.n = -hit.n;
        hit.gn = normalize(hit.n + 0.2f * unit()); // (a geometric normal near the shading normal: the strict-normals test has something to reject)
        ortho_basis(hit.v_x, hit.v_y, hit.n);
        hit.primitiveId = hit.instanceId = 0;
        glm::vec3 wo = unit();
        if (dot(wo, hit.n) < 0.0f) wo = -wo;
        const glm::vec2 u_dir(U(gen), U(gen)), u_sel(U(gen), U(gen));
        binned::NEEQueryAux aux;
        aux.light_dir = glm::vec3(0.0f);
        aux.light_dist = 0.0f;
        aux.mis_pdf = 0.0f;
        const glm::vec3 L = binned::sample_direct_light(m, hit, wo, u_dir, u_sel, aux);
        std::printf("{");
        p3("base_color", m.base_color);
        std::printf("\"metallic\": %.9g, \"specular\": %.9g, \"roughness\": %.9g, \"ior\": %.9g, ", m.metallic, m.specular, m.roughness, m.ior);
        p3("p", hit.p); p3("n", hit.n); p3("gn", hit.gn); p3("wo", wo);
        std::printf("\"u\": [%.9g, %.9g, %.9g, %.9g], ", u_dir.x, u_dir.y, u_sel.x, u_sel.y);
        p3("illum", L);
        std::printf("\"mis_pdf\": %.9g}%s\n", aux.mis_pdf, i + 1 < n_nee ? "," : "");
    }
    std::printf("]},\n\"approx_tri_lights_pdf\": [");
    bin_size = 16;
    for (int i = 0; i < 16; ++i) {
        const float sa = 1e-4f * float(1 << i);
        std::printf("[%.9g, %.9g]%s", sa, binned::approx_tri_lights_pdf(sa), i < 15 ? ", " : "");
    }
    std::printf("]\n}\n");
    return 0;
}
