// ref_shade_driver.cpp -- TEST INFRASTRUCTURE, built only where a real GLM is present (`make -C oracle ref_shaders GLM_ROOT=<dir holding glm/glm.hpp>`).
//
// SURVEY.md section 8(c) group (9) whole: one call of the megakernel's shading step, shade_megakernel -> shade_base_material
// (rendering/mc/shade_megakernel.glsl:13-45, rendering/mc/shade_base_material.glsl:14-96), compiled as C++ from the reference's files WHERE
// THEY LIE, in the include order of vulkan/pt_megakernel.glsl:18-110 with the compile-time features of librender/render_params.glsl.h:7-13
// (USE_MIPMAPPING, UNROLL_STANDARD_TEXTURES) and vulkan/gpu_params.glsl:12 (PREMULTIPLIED_BASE_COLOR_ALPHA):
//     unpack_material (rt/material_textures.glsl:95-135, the unrolled standard-texture path) -> direct emitter hit with its MIS weight ->
//     path-depth cut -> sample_direct_light (mc/nee.glsl) -> glossy-only cut -> sample_bsdf (bsdfs/gltf_bsdf.glsl) -> termination tests ->
//     throughput and prev_bounce_pdf, and the generator's state after the step (the order of the draws is part of the vector).
// ref_shader_driver.cpp is a separate program because it includes gltf_bsdf.glsl WITHOUT material registration and simple_bsdf.glsl (which
// defines SIMPLIFIED_MATERIAL): this one needs the megakernel's macro state instead.
//
// What stands in for Vulkan objects here (none of it is a reference header; all of it is in this file):
//   * a texture is ONE texel (`Texel1x1`): textureGrad / textureLod of a 1 x 1 image return that texel under any filter, which is how the
//     reference's loader represents a constant parameter (librender/scene.cpp: the 1x1 default textures). The three standard textures of
//     material m are standard_textures[3 m + slot] (render_vulkan.cpp:1770-1797).
//   * raytrace_test_visibility answers "visible" (rendering/tests/compile.cpp:39 does the same): the oracle side of the test moves the
//     occluders away.
//   * scene_params / view_params / render_params are plain structs of the reference's own types (vulkan/gpu_params.glsl:56-131 is GLSL-only
//     only because of its layout qualifiers; the members used by the included code are declared here with the same names and types).
// Output: tests/golden/ref_shade.json, read by tests/test_ref_shaders.py. Until the file exists that test skips.
//
// Three things the reference's GLSL asks of a C++ compiler here that its own compile test (rendering/tests/compile.cpp) never needed:
//   * mc/shading_interface.glsl:21 `ShadingSampleState(0, 0, 2.e16f)` initialises an aggregate with parentheses: -std=c++20.
//   * mc/shade_base_material.glsl:64 swizzles a vec4 (`.xy`, `.zw`): GLM offers member swizzles with -DGLM_FORCE_SWIZZLE where anonymous structs are
//     on (clang: -fms-extensions). The recipe passes both.
//   * mc/shade_base_material.glsl:61 `vec4(RANDOM_FLOAT2(..), RANDOM_FLOAT2(..))` makes two draws inside ONE constructor call. GLSL evaluates arguments left to
//     right (position sample first, then light selection); C++ leaves the order open and gcc goes right to left, which swaps the two samples (seen
//     here: the generator's end state agrees, the NEE term of 2 in 3 vectors does not). The recipe compiles this file with clang++, which goes left
//     to right, and main() REFUSES to write vectors when a probe of the same shape says otherwise.
// NOTE for whoever runs this first: it has never been compiled against a real GLM (the build container has none). If another of the reference's
// headers needs a hook, add it HERE, not in a stand-in header.
#include <glm/glm.hpp>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

namespace ref_shade {

using namespace glm;
#include "rendering/language.hpp"

#define PREMULTIPLIED_BASE_COLOR_ALPHA // vulkan/gpu_params.glsl:12
#include "librender/render_params.glsl.h" // USE_MIPMAPPING, UNROLL_STANDARD_TEXTURES, RenderParams, LightSamplingConfig, GLOSSY_MODE_ROUGHNESS_THRESHOLD

#include "rendering/pathspace.h"
#include "rendering/pointsets/lcg_rng.glsl" // RBO_rng_variant = RNG_VARIANT_UNIFORM (the default): defaults.glsl maps RANDOM_* onto LCGRand

#include "rendering/rt/materials.glsl" // MATERIAL_PARAMS = BaseMaterial, EmitterInteraction, the unrolled standard-parameter macros
#include "rendering/bsdfs/gltf_bsdf.glsl" // registers MATERIAL_TYPE / eval_bsdf / eval_bsdf_wpdf / sample_bsdf, load_material

#include "rendering/lights/tri.glsl"

// ---- what the megakernel binds (vulkan/pt_megakernel.glsl:49-103), as host memory
struct Texel1x1 {
    vec4 texel;
};
inline vec4 textureGrad(const Texel1x1 &t, vec2, vec2, vec2) { return t.texel; }
inline vec4 textureLod(const Texel1x1 &t, vec2, float) { return t.texel; }

static const int driver_num_materials = 64;
static Texel1x1 standard_textures[driver_num_materials * STANDARD_TEXTURE_COUNT];
static Texel1x1 textures[1];

static const int driver_num_lights = 40;
static TriLightData global_lights[driver_num_lights + 16] = {}; // padded with a zeroed bin, as in ref_shader_driver.cpp

struct LightSamplingSceneParamsOfTheDriver { // vulkan/gpu_params.glsl:113-118
    int32_t light_count;
};
struct SceneParamsOfTheDriver { // vulkan/gpu_params.glsl:120-131, the members the included code reads
    vec3 sun_dir;
    float sun_cos_angle;
    vec4 sun_radiance;
    LightSamplingSceneParamsOfTheDriver light_sampling;
};
struct ViewParamsOfTheDriver { // vulkan/gpu_params.glsl:56-83
    LightSamplingConfig light_sampling;
};
static SceneParamsOfTheDriver scene_params;
static ViewParamsOfTheDriver view_params;
static RenderParams render_params;

#define SCENE_GET_TEXTURE(tex_id) textures[tex_id]
#define SCENE_GET_STANDARD_TEXTURE(tex_id) standard_textures[tex_id]
#define SCENE_GET_LIGHT_SOURCE(light_id) decode_tri_light(global_lights[light_id])
#define SCENE_GET_LIGHT_SOURCE_COUNT() int(scene_params.light_sampling.light_count)
#define BINNED_LIGHTS_BIN_SIZE int(view_params.light_sampling.bin_size)
#define SCENE_GET_BINNED_LIGHTS_BIN_COUNT() \
    (int(scene_params.light_sampling.light_count + (view_params.light_sampling.bin_size - 1)) / int(view_params.light_sampling.bin_size))

#define CUSTOM_MATERIAL_ALPHA
#include "rendering/rt/material_textures.glsl"
#include "rendering/mc/nee.glsl"
#include "rendering/mc/shade_megakernel.glsl"

inline bool raytrace_test_visibility(const vec3 from, const vec3 dir, float dist) { return true; }

} // namespace ref_shade

// the shape of shade_base_material.glsl:61: two side effects inside one constructor call
static int order_probe_counter = 0;
static int order_probe_next() { return order_probe_counter++; }
static bool call_arguments_left_to_right() {
    order_probe_counter = 0;
    const glm::ivec2 v(order_probe_next(), order_probe_next());
    return v.x == 0 && v.y == 1;
}

static void p3(const char *k, const glm::vec3 &v, bool last = false) { std::printf("\"%s\": [%.9g, %.9g, %.9g]%s", k, v.x, v.y, v.z, last ? "" : ", "); }

int main() {
    using namespace ref_shade;
    std::mt19937 gen(20240921u);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    if (!call_arguments_left_to_right()) {
        std::fprintf(stderr, "ref_shade_driver: this compiler evaluates call arguments right to left; GLSL goes left to right and shade_base_material.glsl:61 "
                             "depends on it. Compile with clang++ (make -C oracle ref_shaders SHADE_CXX=clang++).\n");
        return 3;
    }
    // (the driver's own inputs are drawn in separate statements: what is printed must not depend on the compiler either)
    auto rnd3 = [&](float scale, float offset) {
        glm::vec3 v;
        v.x = scale * U(gen) + offset;
        v.y = scale * U(gen) + offset;
        v.z = scale * U(gen) + offset;
        return v;
    };
    auto unit = [&]() {
        glm::vec3 v;
        do v = rnd3(2.0f, -1.0f);
        while (glm::dot(v, v) < 1e-3f || glm::dot(v, v) > 1.0f);
        return glm::normalize(v);
    };

    // the light table of ref_shader_driver.cpp's "nee" group, made the same way from this file's own seed
    for (int i = 0; i < driver_num_lights; ++i) {
        glm::vec3 c = rnd3(1.0f, 0.0f) * glm::vec3(8.0f, 2.0f, 8.0f) + glm::vec3(-4.0f, 2.0f, -4.0f);
        glm::vec3 e1 = 0.5f * unit();
        glm::vec3 e2 = 0.5f * unit();
        TriLightData &l = global_lights[i];
        std::memset(&l, 0, sizeof(l));
        l.v0_x = c.x, l.v0_y = c.y, l.v0_z = c.z;
        l.v1_x = c.x + e1.x, l.v1_y = c.y + e1.y, l.v1_z = c.z + e1.z;
        l.v2_x = c.x + e2.x, l.v2_y = c.y + e2.y, l.v2_z = c.z + e2.z;
        const glm::vec3 L = rnd3(9.0f, 1.0f);
        l.radiance_x = L.x, l.radiance_y = L.y, l.radiance_z = L.z;
    }
    scene_params.light_sampling.light_count = driver_num_lights;
    view_params.light_sampling = LightSamplingConfig();
    scene_params.sun_dir = glm::normalize(glm::vec3(0.3f, 0.8f, 0.5f));
    scene_params.sun_cos_angle = std::cos(0.00465f * 4);
    scene_params.sun_radiance = glm::vec4(12.0f, 11.0f, 9.0f, 0.25f);
    render_params = RenderParams();

    // materials: the literal members of BaseMaterial + the texels of its three standard textures. base colour, roughness and metallic are
    // read from the texels ALWAYS (material_textures.glsl:99,107-115), specular and ior from the struct unless flagged as handles (never, here)
    static BaseMaterial materials[driver_num_materials];
    const float rough_set[8] = {0.0f, 0.02f, 0.1f, 0.25f, 0.5f, 0.75f, 0.9f, 1.0f};
    for (int m = 0; m < driver_num_materials; ++m) {
        BaseMaterial &p = materials[m];
        p = BaseMaterial();
        const bool emitter = (m % 8) == 7;
        glm::vec3 col = rnd3(0.9f, 0.05f);
        float rough = rough_set[m % 8 == 7 ? 4 : (m % 8)], metal = (m / 8) % 2 ? 1.0f : 0.0f;
        if ((m / 16) % 2) metal = U(gen);
        p.base_color = col;
        p.roughness = rough;
        p.metallic = metal;
        p.specular = (m / 32) ? U(gen) : 0.0f;
        p.ior = 1.3f + 0.4f * U(gen);
        p.emission_intensity = emitter ? 1.0f + 20.0f * U(gen) : 0.0f;
        standard_textures[STANDARD_TEXTURE_COUNT * m + STANDARD_TEXTURE_BASECOLOR_SLOT].texel = glm::vec4(col, 1.0f);
        standard_textures[STANDARD_TEXTURE_COUNT * m + STANDARD_TEXTURE_SPECULAR_SLOT].texel = glm::vec4(p.specular, rough, metal, 0.0f);
        standard_textures[STANDARD_TEXTURE_COUNT * m + STANDARD_TEXTURE_NORMAL_SLOT].texel = glm::vec4(0.5f, 0.5f, 1.0f, 1.0f);
    }

    std::printf("{\n\"lights\": [");
    for (int i = 0; i < driver_num_lights; ++i) {
        const TriLightData &l = global_lights[i];
        std::printf("%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", l.v0_x, l.v0_y, l.v0_z, l.v1_x, l.v1_y, l.v1_z,
                    l.v2_x, l.v2_y, l.v2_z, l.radiance_x, l.radiance_y, l.radiance_z);
    }
    std::printf("],\n\"bin_size\": %d, \"light_mis_angle\": %.9g, \"min_perceived_receiver_dist\": %.9g, \"min_radiance\": %.9g,\n", view_params.light_sampling.bin_size,
                view_params.light_sampling.light_mis_angle, view_params.light_sampling.min_perceived_receiver_dist, view_params.light_sampling.min_radiance);
    std::printf("\"sun_dir\": [%.9g, %.9g, %.9g], \"sun_cos_angle\": %.9g, \"sun_radiance\": [%.9g, %.9g, %.9g, %.9g],\n", scene_params.sun_dir.x, scene_params.sun_dir.y,
                scene_params.sun_dir.z, scene_params.sun_cos_angle, scene_params.sun_radiance.x, scene_params.sun_radiance.y, scene_params.sun_radiance.z,
                scene_params.sun_radiance.w);
    std::printf("\"materials\": [");
    for (int m = 0; m < driver_num_materials; ++m) {
        const BaseMaterial &p = materials[m];
        std::printf("%s{\"base_color\": [%.9g, %.9g, %.9g], \"roughness\": %.9g, \"metallic\": %.9g, \"specular\": %.9g, \"ior\": %.9g, \"emission_intensity\": %.9g}",
                    m ? ",\n " : "", p.base_color.x, p.base_color.y, p.base_color.z, p.roughness, p.metallic, p.specular, p.ior, p.emission_intensity);
    }
    std::printf("],\n\"samples\": [\n");

    const int N = 512;
    for (int i = 0; i < N; ++i) {
        const int material_id = int(gen() % driver_num_materials);
        InteractionPoint ip;
        ip.p = rnd3(1.0f, -0.5f) * glm::vec3(6.0f, 1.0f, 6.0f);
        ip.gn = (i % 4 == 3) ? unit() : glm::normalize(rnd3(0.3f, -0.15f) * glm::vec3(1.0f, 0.0f, 1.0f) + glm::vec3(0.0f, 1.0f, 0.0f));
        // a shading normal near the geometric one, and its frame (v_x, v_y) as the megakernel passes it: orthonormal, right-handed about n
        ip.n = glm::normalize(ip.gn + 0.35f * unit());
        glm::vec3 t = glm::normalize(glm::cross(ip.n, std::fabs(ip.n.x) < 0.9f ? glm::vec3(1, 0, 0) : glm::vec3(0, 1, 0)));
        ip.v_x = t;
        ip.v_y = glm::cross(ip.n, t);
        ip.primitiveId = 0;
        ip.instanceId = 0;
        glm::vec3 w_o = unit();
        if (i % 8 != 5 && glm::dot(w_o, ip.n) < 0.0f) w_o = -w_o; // one in eight from below: the BSDF's own hemisphere tests
        HitPoint lookup;
        lookup.p = ip.p;
        lookup.uv.x = U(gen);
        lookup.uv.y = U(gen);
        lookup.duvdxy = glm::mat2(0.0f);
        lookup.d = -w_o;
        ShadingSampleState state = init_shading_sample_state();
        state.bounce = int(gen() % 4);
        if (i % 16 == 9) state.bounce = render_params.max_path_depth - 1; // the path-depth cut
        state.output_channel = (i % 32 == 17) ? 1 + int(gen() % 3) : 0;
        state.prev_bounce_pdf = state.bounce == 0 ? 2.e16f : 0.05f + 4.0f * U(gen);
        render_params.glossy_only_mode = (i % 32 == 21) ? 1 : 0;
        NEESampledArea area;
        area.approx_solid_angle = 0.001f + 0.2f * U(gen);
        area.type = 0;
        glm::vec3 illum = rnd3(1.0f, 0.0f);
        glm::vec3 throughput = rnd3(1.0f, 0.1f);
        LCGRand rng;
        rng.state = gen();
        const ShadingSampleState state_in = state;
        const glm::vec3 illum_in = illum, throughput_in = throughput;
        const uint32_t rng_in = rng.state;
        glm::vec3 w_i(0.0f);
        ShadingQueryAux aux;
        aux.sampling_pdf = 0.0f;
        aux.mis_pdf = 0.0f;
        int result = shade_megakernel(state, illum, throughput, material_id, materials[material_id], lookup, area, w_o, ip, rng, w_i, aux, true);
        const bool bounce = result == SHADING_RESULT_BOUNCE;
        std::printf(" {\"material\": %d, ", material_id);
        p3("p", ip.p), p3("gn", ip.gn), p3("n", ip.n), p3("v_x", ip.v_x), p3("v_y", ip.v_y), p3("w_o", w_o);
        std::printf("\"bounce\": %d, \"output_channel\": %d, \"prev_bounce_pdf\": %.9g, \"glossy_only_mode\": %d, \"approx_solid_angle\": %.9g, \"rng\": %u, ", state_in.bounce,
                    state_in.output_channel, state_in.prev_bounce_pdf, render_params.glossy_only_mode, area.approx_solid_angle, rng_in);
        p3("illum_in", illum_in), p3("throughput_in", throughput_in);
        std::printf("\"result\": %d, \"bounce_out\": %d, \"rng_out\": %u, ", result, state.bounce, rng.state);
        p3("illum", illum);
        if (bounce) {
            p3("w_i", w_i), p3("throughput", throughput);
            std::printf("\"prev_bounce_pdf_out\": %.9g", state.prev_bounce_pdf);
        } else
            std::printf("\"terminated\": true"); // w_i / throughput / pdf are not defined after a terminating step (shade_base_material.glsl:87-89)
        std::printf("}%s\n", i + 1 < N ? "," : "");
    }
    std::printf("]\n}\n");
    return 0;
}
