// TEST INFRASTRUCTURE (oracle/_ref/libabi_ref.so, built by oracle/Makefile where the reference checkout is): the reference's own
// dual-language struct / constant headers compiled where they lie, so that the tests can hold include/rptr_hip.h and abi.py against
// them field by field. These headers are written for C++ AND GLSL: vector members are spelled GLM(type) and the headers only pull in
// glm when nobody defined that macro (librender/render_params.glsl.h:22-25). This file defines it to plain float aggregates of the
// sizes the std430 / scalar layouts give the GLSL side -- nothing of glm's interface is imitated, no arithmetic is done on them.
// The one exported function prints sizes, member offsets, default values and constants as JSON.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <string>

struct abi_vec2 { float v[2]; };
struct abi_vec3 {
    float v[3];
    abi_vec3() : v{0, 0, 0} {}
    explicit abi_vec3(float a) : v{a, a, a} {}
    abi_vec3(float a, float b, float c) : v{a, b, c} {}
};
struct abi_vec4 { float v[4]; };
#define GLM(type) abi_##type

#include "librender/render_params.glsl.h"
#include "rendering/bsdfs/base_material.h.glsl"
#include "rendering/bsdfs/texture_channel_mask.h"
#include "rendering/lights/sky_model_arhosek/sky_model.h.glsl"
#include "rendering/lights/tri.h.glsl"
#include "rendering/pathspace.h"
#include "rendering/pointsets/bn_data.h"
#include "rendering/pointsets/sobol_data.h"
#include "librender/halton.h" // the (2, 3) Halton table behind view_params.screen_jitter (vulkan/render_vulkan.cpp:2917-2926)

static std::string g_json;
static void put(const char *k, double v) {
    char b[160];
    snprintf(b, sizeof(b), "%s\"%s\": %.17g", g_json.size() > 1 ? ", " : "", k, v);
    g_json += b;
}
#define SIZE(T) put("sizeof_" #T, (double)sizeof(T))
#define OFF(T, m) put("offsetof_" #T "_" #m, (double)offsetof(T, m))
#define DEF(T, obj, m) put("default_" #T "_" #m, (double)obj.m)
#define FIELD(T, obj, m) OFF(T, m), DEF(T, obj, m)
#define CONST(name) put(#name, (double)(name))

extern "C" const char *ref_abi_json() {
    g_json = "{";
    RenderParams rp;
    SIZE(RenderParams);
    FIELD(RenderParams, rp, batch_spp); FIELD(RenderParams, rp, max_path_depth); FIELD(RenderParams, rp, rr_path_depth); FIELD(RenderParams, rp, glossy_only_mode);
    FIELD(RenderParams, rp, aperture_radius); FIELD(RenderParams, rp, focus_distance); FIELD(RenderParams, rp, pixel_radius); FIELD(RenderParams, rp, variance_radius);
    FIELD(RenderParams, rp, output_channel); FIELD(RenderParams, rp, output_moment); FIELD(RenderParams, rp, exposure); FIELD(RenderParams, rp, early_tone_mapping_mode);
    FIELD(RenderParams, rp, reprojection_mode); FIELD(RenderParams, rp, spp_accumulation_window); FIELD(RenderParams, rp, enable_raster_taa);
    FIELD(RenderParams, rp, render_upscale_factor); FIELD(RenderParams, rp, focal_length);
    LightSamplingConfig lc;
    SIZE(LightSamplingConfig);
    FIELD(LightSamplingConfig, lc, light_mis_angle); FIELD(LightSamplingConfig, lc, bin_size); FIELD(LightSamplingConfig, lc, min_perceived_receiver_dist);
    FIELD(LightSamplingConfig, lc, min_radiance);
    RenderBackendOptions bo;
    DEF(RenderBackendOptions, bo, rng_variant); DEF(RenderBackendOptions, bo, light_sampling_variant); DEF(RenderBackendOptions, bo, light_sampling_bucket_count);
    DEF(RenderBackendOptions, bo, render_upscale_factor); DEF(RenderBackendOptions, bo, force_bvh_rebuild); DEF(RenderBackendOptions, bo, rebuild_triangle_budget);
    SceneConfig sc;
    DEF(SceneConfig, sc, bump_scale); DEF(SceneConfig, sc, turbidity);
    put("default_SceneConfig_sun_dir_y", sc.sun_dir.v[1]);
    put("default_SceneConfig_albedo_x", sc.albedo.v[0]);
    SIZE(RenderRayQuery);
    OFF(RenderRayQuery, origin); OFF(RenderRayQuery, mode_or_data); OFF(RenderRayQuery, dir); OFF(RenderRayQuery, t_max);
    BaseMaterial bm;
    SIZE(BaseMaterial);
    OFF(BaseMaterial, base_color); FIELD(BaseMaterial, bm, normal_map); FIELD(BaseMaterial, bm, flags); FIELD(BaseMaterial, bm, roughness); FIELD(BaseMaterial, bm, specular);
    FIELD(BaseMaterial, bm, metallic); FIELD(BaseMaterial, bm, sheen); FIELD(BaseMaterial, bm, sheen_tint); FIELD(BaseMaterial, bm, clearcoat);
    FIELD(BaseMaterial, bm, clearcoat_gloss); FIELD(BaseMaterial, bm, ior); FIELD(BaseMaterial, bm, specular_transmission); FIELD(BaseMaterial, bm, anisotropy);
    FIELD(BaseMaterial, bm, specular_tint); OFF(BaseMaterial, transmission_color); FIELD(BaseMaterial, bm, emission_intensity);
    put("default_BaseMaterial_base_color_x", bm.base_color.v[0]);
    put("default_BaseMaterial_transmission_color_x", bm.transmission_color.v[0]);
    SIZE(TriLightData);
    OFF(TriLightData, v0_x); OFF(TriLightData, v1_x); OFF(TriLightData, v2_x); OFF(TriLightData, radiance_x);
    SIZE(SkyModelParams);
    OFF(SkyModelParams, configs); OFF(SkyModelParams, radiances);
    SIZE(SobolData);
    OFF(SobolData, matrix); OFF(SobolData, tile_invert_1_0);
    SIZE(BNData);
    OFF(BNData, sobol_spp_d); OFF(BNData, tile_scrambling_yx_d_1spp); OFF(BNData, tile_scrambling_yx_d_4spp);
    CONST(SobolData_Dimensions); CONST(SobolData_MatrixSize); CONST(SobolData_TileSize); CONST(BNData_SampleCount); CONST(BNData_Dimensions);
    CONST(BNData_ScramblingDimensions); CONST(BNData_TileSize);
    CONST(MAX_PATH_DEPTH); CONST(DEFAULT_RR_PATH_DEPTH); CONST(BINNED_LIGHTS_BIN_MAX_SIZE); CONST(GLOSSY_MODE_ROUGHNESS_THRESHOLD); CONST(DEFAULT_RAY_QUERY_BUDGET);
    CONST(RNG_VARIANT_UNIFORM); CONST(RNG_VARIANT_BN); CONST(RNG_VARIANT_SOBOL); CONST(RNG_VARIANT_Z_SBL);
    CONST(OUTPUT_CHANNEL_COLOR); CONST(OUTPUT_CHANNEL_ALBEDO_ROUGHNESS); CONST(OUTPUT_CHANNEL_NORMAL_DEPTH); CONST(OUTPUT_CHANNEL_MOTION_JITTER);
    CONST(REPROJECTION_MODE_NONE); CONST(REPROJECTION_MODE_DISCARD_HISTORY); CONST(REPROJECTION_MODE_ACCUMULATE);
    CONST(LIGHT_SAMPLING_VARIANT_NONE); CONST(LIGHT_SAMPLING_VARIANT_RIS);
    CONST(BASE_MATERIAL_NOALPHA); CONST(BASE_MATERIAL_ONESIDED); CONST(BASE_MATERIAL_VOLUME); CONST(BASE_MATERIAL_EXTENDED);
    CONST(STANDARD_TEXTURE_COUNT); CONST(STANDARD_TEXTURE_BASECOLOR_SLOT); CONST(STANDARD_TEXTURE_NORMAL_SLOT); CONST(STANDARD_TEXTURE_SPECULAR_SLOT);
    CONST(DIM_PIXEL_X); CONST(DIM_PIXEL_Y); CONST(DIM_APERTURE_X); CONST(DIM_CAMERA_END); CONST(DIM_DIRECTION_X); CONST(DIM_DIRECTION_Y); CONST(DIM_LOBE);
    CONST(DIM_FREE_PATH); CONST(DIM_VERTEX_END); CONST(DIM_RR); CONST(DIM_LIGHT_SEL_1); CONST(DIM_LIGHT_SEL_2); CONST(DIM_POSITION_X); CONST(DIM_POSITION_Y);
    CONST(DIM_LIGHT_END);
    {   // a texture handle through the reference's macros: texture 1234, channel 2
        uint32_t h = TEXTURED_PARAM_MASK;
        SET_TEXTURE_ID(h, 1234);
        SET_TEXTURE_CHANNEL(h, 2);
        put("texture_handle_1234_2", (double)h);
        put("texture_handle_id", (double)GET_TEXTURE_ID(h));
        put("texture_handle_channel", (double)GET_TEXTURE_CHANNEL(h));
    }
    for (int k = 0; k < 16; ++k) { // the 16 entries the jitter cycles through
        char key[32];
        snprintf(key, sizeof(key), "halton_23_%d_x", k);
        put(key, (double)halton_23[k][0]);
        snprintf(key, sizeof(key), "halton_23_%d_y", k);
        put(key, (double)halton_23[k][1]);
    }
    g_json += "}";
    return g_json.c_str();
}
