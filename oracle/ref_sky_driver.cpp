// TEST INFRASTRUCTURE -- built only where /root/reference exists (Makefile
// target `ref`), output oracle/_ref/libsky_ref.so.
//
// Links the reference's Hosek-Wilkie implementation
// (rendering/lights/sky_model_arhosek/sky_model.cpp, compiled where it lies, no
// stand-in headers needed: it depends on libc/libm only) and restates the ~45
// lines of RenderVulkan::update_sky_light (vulkan/render_sky.cpp:25-72) around
// it, so that SkyModelParams / sun_radiance golden vectors for the synthetic
// configs come out of the reference's own fitting code and data tables.
// The CIE table is read from the reference header by path (data only).
#include <cmath>
#include <cstring>

#include "lights/sky_model_arhosek/sky_model.h" // -I$(REF)/rendering
#include "color/color_matching.h"               // cie1931_tbl, CM_CIE_*

extern "C" {

struct RefSkyOut {
    float configs[9][4];
    float radiances[4];
    float sun_dir[3];
    float sun_cos_angle;
    float sun_radiance[4];
};

// render_sky.cpp:25-72. light_count: scene_params.light_sampling.light_count
int ref_update_sky_light(const float sun_dir_in[3], float turbidity, const float albedo[3], int light_count, RefSkyOut *out) {
    float l = std::sqrt(sun_dir_in[0] * sun_dir_in[0] + sun_dir_in[1] * sun_dir_in[1] + sun_dir_in[2] * sun_dir_in[2]);
    float inv = 1.0f / l; // glm::normalize = v * inversesqrt(dot(v,v))
    float sun_dir[3] = {sun_dir_in[0] * inv, sun_dir_in[1] * inv, sun_dir_in[2] * inv};

    ArHosekSkyModelState state;
    float albedo_avg = albedo[0] * 0.3333f + albedo[1] * 0.3333f + albedo[2] * 0.3333f; // dot(albedo, vec3(0.3333))
    arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo_avg, sun_dir[1], &state);

    memset(out, 0, sizeof(*out));
    memcpy(out->sun_dir, sun_dir, sizeof(sun_dir));
    out->sun_cos_angle = std::cos((0.53f * 0.01745329251994329576923690768489f) / 2.0f);
    for (int i = 0; i < 9; ++i) {
        out->configs[i][0] = (float)state.configs[0][i];
        out->configs[i][1] = (float)state.configs[1][i];
        out->configs[i][2] = (float)state.configs[2][i];
        out->configs[i][3] = 0.0f;
    }
    out->radiances[0] = (float)state.radiances[0];
    out->radiances[1] = (float)state.radiances[1];
    out->radiances[2] = (float)state.radiances[2];

    ArHosekSkyModelState sunState;
    arhosekskymodelstate_alloc_init(state.elevation, state.turbidity, state.albedo, &sunState);
    float xyz[3] = {0, 0, 0};
    int numSamples = 0;
    float last_wavelength = CM_CIE_MIN;
    const float *TX = &cie1931_tbl[0], *TY = &cie1931_tbl[CM_CIE_SAMPLES], *TZ = &cie1931_tbl[2 * CM_CIE_SAMPLES];
    for (int i = 0; i < CM_CIE_SAMPLES; ++i) {
        float wavelength = float(i) * float(CM_CIE_MAX - CM_CIE_MIN) / float(CM_CIE_SAMPLES - 1) + float(CM_CIE_MIN);
        if (wavelength > 720.0f) break;
        float radiance = arhosekskymodel_solar_radiance(&sunState, sun_dir[1], 0.0, wavelength);
        radiance -= arhosekskymodel_radiance(&sunState, sun_dir[1], 0.0, wavelength); // float -= double, as render_sky.cpp:52 writes it
        xyz[0] += TX[i] * radiance;
        xyz[1] += TY[i] * radiance;
        xyz[2] += TZ[i] * radiance;
        ++numSamples;
        last_wavelength = wavelength;
    }
    float scale = float(last_wavelength - CM_CIE_MIN) / float(numSamples);
    for (int k = 0; k < 3; ++k) xyz[k] *= scale;
    // xyz_to_srgb, rendering/color/color_matching.glsl:87-92 (M = transpose(mat3(rows...)) => rows below)
    const float M[3][3] = {{3.240479f, -1.537150f, -0.498535f}, {-0.969256f, 1.875991f, 0.041556f}, {0.055648f, -0.204043f, 1.057311f}};
    if (sun_dir[1] > 0.0f && xyz[0] >= 0.0f && xyz[1] >= 0.0f && xyz[2] >= 0.0f) {
        for (int r = 0; r < 3; ++r) out->sun_radiance[r] = 0.01f * ((M[r][0] * xyz[0] + M[r][1] * xyz[1]) + M[r][2] * xyz[2]);
        out->sun_radiance[3] = 1.0f;
    }
    if (light_count > 0)
        out->sun_radiance[3] *= 0.5f;
    else
        out->sun_radiance[3] = 1.0f;
    return 0;
}

// Reference evaluation of the RGB sky (sky_model.cpp:644-658): used to pin the
// GLSL restatement of skymodel_radiance for sun-at-zenith configurations, where
// the shader's gamma (it uses the zenith angle, sky_model.glsl:48) coincides
// with the sun angle the C code expects.
int ref_sky_radiance(const float sun_dir_in[3], float turbidity, const float albedo[3], const double *theta, const double *gamma, int n,
                     double *out_rgb) {
    float l = std::sqrt(sun_dir_in[0] * sun_dir_in[0] + sun_dir_in[1] * sun_dir_in[1] + sun_dir_in[2] * sun_dir_in[2]);
    float sun_y = sun_dir_in[1] * (1.0f / l);
    ArHosekSkyModelState state;
    float albedo_avg = albedo[0] * 0.3333f + albedo[1] * 0.3333f + albedo[2] * 0.3333f;
    arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo_avg, sun_y, &state);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) out_rgb[3 * i + c] = arhosek_tristim_skymodel_radiance(&state, theta[i], gamma[i], c);
    return 0;
}

} // extern "C"
