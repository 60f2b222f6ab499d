// sky_fit.hpp -- the host half of row a15: the Hosek-Wilkie sky fit and the sun's radiance for a scene state (sun direction, turbidity,
// ground albedo) -> RptrSceneParams. Restates RenderVulkan::update_sky_light (vulkan/render_sky.cpp:25-72) and the functions of
// rendering/lights/sky_model_arhosek/sky_model.cpp it calls:
//   arhosek_rgb_skymodelstate_alloc_init :608-642, arhosekskymodelstate_alloc_init :311-348,
//   ArHosekSkyModel_CookConfiguration :150-230, ArHosekSkyModel_CookRadianceConfiguration :232-292,
//   ArHosekSkyModel_GetRadianceInternal :294-307, arhosekskymodel_radiance :524-566,
//   arhosekskymodel_sr_internal :663-692, arhosekskymodel_solar_radiance_internal2 :694-796, arhosekskymodel_solar_radiance :798-822
// and rendering/color/color_matching.glsl:87-92 (xyz_to_srgb).
//
// The model (Hosek & Wilkie, "An Analytic Model for Full Spectral Sky-Dome Radiance", SIGGRAPH 2012; "Adding a Solar-Radiance Function
// to the Hosek-Wilkie Skylight Model", IEEE CG&A 2013) is a fit whose coefficients are 580 KB of published data. This repository does
// not carry them: they are READ AT RUN TIME from the files an integration points at (`--sky-data <dir>` / RPTR_SKY_DATA) -- the
// model's own data headers as distributed by its authors (ArHosekSkyModelData_RGB.h, ArHosekSkyModelData_Spectral.h) or as the
// reference ships them (rendering/lights/sky_model_arhosek/sky_model_data_rgb.h, sky_model_data_spectral.h), and the CIE 1931 table of
// rendering/color/color_matching.h. The headers are parsed as data (`double name[] = { numbers };`), nothing of them is compiled in.
//
// Arithmetic: double for the model (as the reference's C code), float where update_sky_light uses float; same operations in the same
// order, so that the result equals the reference's own code bit for bit (tests/test_sky_fit.py against oracle/_ref/libsky_ref.so).
#pragma once
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/rptr_hip.h"

namespace rptr {

struct SkyTables {
    std::vector<double> rgb[3], rgb_rad[3];                        // datasetRGB1..3 (9 x 6 x 10 x 2), datasetRGBRad1..3 (6 x 10 x 2)
    std::vector<double> spec[11], spec_rad[11], solar[11], limb[11]; // dataset320..720, datasetRad*, solarDataset* (4 x 45 x 10), limbDarkeningDataset* (6)
    std::vector<float> cie;                                        // cie1931_tbl: X[95], Y[95], Z[95]
    bool has_rgb() const { return rgb[0].size() >= 1080 && rgb[1].size() >= 1080 && rgb[2].size() >= 1080 && rgb_rad[0].size() >= 120 && rgb_rad[1].size() >= 120 && rgb_rad[2].size() >= 120; }
    bool has_sun() const {
        for (int w = 0; w < 11; ++w)
            if (spec[w].size() < 1080 || spec_rad[w].size() < 120 || solar[w].size() < 1800 || limb[w].size() < 6) return false;
        return cie.size() >= 3 * 95;
    }
};

// every `type name[...] = { number, number, ... };` of a C header as name -> numbers (comments skipped; `Float(x)` wrappers and
// suffixes like 1.0f accepted)
inline bool parse_c_arrays(const std::string &path, std::map<std::string, std::vector<double>> &out, std::string &err) {
    std::FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::string text;
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    std::fclose(f);
    // strip comments
    std::string s;
    s.reserve(text.size());
    for (size_t i = 0; i < text.size();) {
        if (text.compare(i, 2, "//") == 0) {
            while (i < text.size() && text[i] != '\n') ++i;
        } else if (text.compare(i, 2, "/*") == 0) {
            const size_t e = text.find("*/", i + 2);
            i = e == std::string::npos ? text.size() : e + 2;
        } else
            s += text[i++];
    }
    size_t at = 0;
    while ((at = s.find('[', at)) != std::string::npos) {
        // the identifier before '['
        size_t e = at;
        while (e > 0 && std::isspace((unsigned char)s[e - 1])) --e;
        size_t b = e;
        while (b > 0 && (std::isalnum((unsigned char)s[b - 1]) || s[b - 1] == '_')) --b;
        const std::string name = s.substr(b, e - b);
        const size_t close = s.find(']', at);
        if (close == std::string::npos) break;
        size_t p = close + 1;
        while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
        if (p >= s.size() || s[p] != '=' || name.empty()) {
            at = close + 1;
            continue;
        }
        ++p;
        while (p < s.size() && std::isspace((unsigned char)s[p])) ++p;
        if (p >= s.size() || s[p] != '{') {
            at = close + 1;
            continue;
        }
        const size_t end = s.find('}', p);
        if (end == std::string::npos) break;
        std::vector<double> values;
        bool numeric = true;
        for (size_t q = p + 1; q < end;) {
            const char c = s[q];
            if (std::isdigit((unsigned char)c) || c == '-' || c == '+' || c == '.') {
                char *stop = nullptr;
                values.push_back(std::strtod(s.c_str() + q, &stop));
                q = (size_t)(stop - s.c_str());
                if (q < end && (s[q] == 'f' || s[q] == 'F')) ++q;
            } else if (std::isalpha((unsigned char)c) || c == '_') { // Float( ... ) wrapper -- or the name of another array (pointer tables)
                size_t w = q;
                while (w < end && (std::isalnum((unsigned char)s[w]) || s[w] == '_')) ++w;
                size_t v = w;
                while (v < end && std::isspace((unsigned char)s[v])) ++v;
                if (v < end && s[v] == '(')
                    q = v + 1;
                else {
                    numeric = false;
                    break;
                }
            } else
                ++q;
        }
        if (numeric && !values.empty()) out[name] = std::move(values);
        at = end + 1;
    }
    return true;
}

// `where`: a directory (or several, ':'-separated) holding the data headers -- the model's distribution, or the reference's
// rendering/ directory (lights/sky_model_arhosek/ and color/ below it are searched too)
inline bool load_sky_tables(const std::string &where, SkyTables &t, std::string &err) {
    std::vector<std::string> dirs;
    for (size_t b = 0; b <= where.size();) {
        const size_t e = where.find(':', b);
        const std::string d = where.substr(b, e == std::string::npos ? std::string::npos : e - b);
        if (!d.empty())
            for (const char *sub : {"", "/lights/sky_model_arhosek", "/color", "/rendering/lights/sky_model_arhosek", "/rendering/color", "/../../color"}) dirs.push_back(d + sub);
        if (e == std::string::npos) break;
        b = e + 1;
    }
    auto find = [&](std::initializer_list<const char *> names) -> std::string {
        for (const std::string &d : dirs)
            for (const char *nm : names) {
                const std::string p = d + "/" + nm;
                if (std::FILE *f = std::fopen(p.c_str(), "rb")) {
                    std::fclose(f);
                    return p;
                }
            }
        return std::string();
    };
    const std::string rgb = find({"sky_model_data_rgb.h", "ArHosekSkyModelData_RGB.h"});
    if (rgb.empty()) {
        err = "no sky_model_data_rgb.h / ArHosekSkyModelData_RGB.h under '" + where + "'";
        return false;
    }
    std::map<std::string, std::vector<double>> a;
    if (!parse_c_arrays(rgb, a, err)) return false;
    for (int c = 0; c < 3; ++c) {
        t.rgb[c] = a["datasetRGB" + std::to_string(c + 1)];
        t.rgb_rad[c] = a["datasetRGBRad" + std::to_string(c + 1)];
    }
    if (!t.has_rgb()) {
        err = rgb + ": datasetRGB1..3 / datasetRGBRad1..3 not found or too short";
        return false;
    }
    // the sun needs the spectral data and the colour-matching table; without them the sky is fitted and the sun stays dark (said in err)
    const std::string spec = find({"sky_model_data_spectral.h", "ArHosekSkyModelData_Spectral.h"});
    const std::string cie = find({"color_matching.h"});
    if (!spec.empty() && !cie.empty()) {
        std::map<std::string, std::vector<double>> s, c;
        if (!parse_c_arrays(spec, s, err) || !parse_c_arrays(cie, c, err)) return false;
        for (int w = 0; w < 11; ++w) {
            const std::string wl = std::to_string(320 + 40 * w);
            t.spec[w] = s["dataset" + wl];
            t.spec_rad[w] = s["datasetRad" + wl];
            t.solar[w] = s["solarDataset" + wl];
            t.limb[w] = s["limbDarkeningDataset" + wl];
        }
        for (double v : c["cie1931_tbl"]) t.cie.push_back((float)v);
        if (!t.has_sun()) {
            err = spec + " / " + cie + ": spectral datasets or cie1931_tbl incomplete";
            return false;
        }
    } else
        err = "sky fitted, sun left dark: " + std::string(spec.empty() ? "sky_model_data_spectral.h" : "color_matching.h") + " not found under '" + where + "'";
    return true;
}

namespace skyfit {

constexpr double PI = 3.141592653589793;

// sky_model.cpp:150-230: quintic Bezier in cbrt(elevation / (pi/2)), bilinear in turbidity and albedo
inline void cook_configuration(const double *dataset, double config[9], double turbidity, double albedo, double solar_elevation) {
    const int int_turbidity = (int)turbidity;
    const double turbidity_rem = turbidity - (double)int_turbidity;
    solar_elevation = std::pow(solar_elevation / (PI / 2.0), (1.0 / 3.0));
    auto bezier = [&](const double *m, unsigned i) {
        return std::pow(1.0 - solar_elevation, 5.0) * m[i] + 5.0 * std::pow(1.0 - solar_elevation, 4.0) * solar_elevation * m[i + 9] +
               10.0 * std::pow(1.0 - solar_elevation, 3.0) * std::pow(solar_elevation, 2.0) * m[i + 18] +
               10.0 * std::pow(1.0 - solar_elevation, 2.0) * std::pow(solar_elevation, 3.0) * m[i + 27] +
               5.0 * (1.0 - solar_elevation) * std::pow(solar_elevation, 4.0) * m[i + 36] + std::pow(solar_elevation, 5.0) * m[i + 45];
    };
    const double *m = dataset + (9 * 6 * (int_turbidity - 1)); // albedo 0, low turbidity
    for (unsigned i = 0; i < 9; ++i) config[i] = (1.0 - albedo) * (1.0 - turbidity_rem) * bezier(m, i);
    m = dataset + (9 * 6 * 10 + 9 * 6 * (int_turbidity - 1)); // albedo 1, low turbidity
    for (unsigned i = 0; i < 9; ++i) config[i] += (albedo) * (1.0 - turbidity_rem) * bezier(m, i);
    if (int_turbidity == 10) return;
    m = dataset + (9 * 6 * (int_turbidity)); // albedo 0, high turbidity
    for (unsigned i = 0; i < 9; ++i) config[i] += (1.0 - albedo) * (turbidity_rem) * bezier(m, i);
    m = dataset + (9 * 6 * 10 + 9 * 6 * (int_turbidity)); // albedo 1, high turbidity
    for (unsigned i = 0; i < 9; ++i) config[i] += (albedo) * (turbidity_rem) * bezier(m, i);
}

// sky_model.cpp:232-292
inline double cook_radiance_configuration(const double *dataset, double turbidity, double albedo, double solar_elevation) {
    const int int_turbidity = (int)turbidity;
    const double turbidity_rem = turbidity - (double)int_turbidity;
    solar_elevation = std::pow(solar_elevation / (PI / 2.0), (1.0 / 3.0));
    auto bezier = [&](const double *m) {
        return std::pow(1.0 - solar_elevation, 5.0) * m[0] + 5.0 * std::pow(1.0 - solar_elevation, 4.0) * solar_elevation * m[1] +
               10.0 * std::pow(1.0 - solar_elevation, 3.0) * std::pow(solar_elevation, 2.0) * m[2] +
               10.0 * std::pow(1.0 - solar_elevation, 2.0) * std::pow(solar_elevation, 3.0) * m[3] +
               5.0 * (1.0 - solar_elevation) * std::pow(solar_elevation, 4.0) * m[4] + std::pow(solar_elevation, 5.0) * m[5];
    };
    double res = (1.0 - albedo) * (1.0 - turbidity_rem) * bezier(dataset + (6 * (int_turbidity - 1)));
    res += (albedo) * (1.0 - turbidity_rem) * bezier(dataset + (6 * 10 + 6 * (int_turbidity - 1)));
    if (int_turbidity == 10) return res;
    res += (1.0 - albedo) * (turbidity_rem) * bezier(dataset + (6 * (int_turbidity)));
    res += (albedo) * (turbidity_rem) * bezier(dataset + (6 * 10 + 6 * (int_turbidity)));
    return res;
}

// sky_model.cpp:294-307
inline double radiance_internal(const double c[9], double theta, double gamma) {
    const double expM = std::exp(c[4] * gamma);
    const double rayM = std::cos(gamma) * std::cos(gamma);
    const double mieM = (1.0 + std::cos(gamma) * std::cos(gamma)) / std::pow((1.0 + c[8] * c[8] - 2.0 * c[8] * std::cos(gamma)), 1.5);
    const double zenith = std::sqrt(std::cos(theta));
    return (1.0 + c[0] * std::exp(c[1] / (std::cos(theta) + 0.01))) * (c[2] + c[3] * expM + c[5] * rayM + c[6] * mieM + c[7] * zenith);
}

struct SpectralState { // ArHosekSkyModelState of the spectral model (sky_model.cpp:311-348)
    double configs[11][9], radiances[11], turbidity, solar_radius, albedo, elevation;
};

// sky_model.cpp:524-566
inline double spectral_radiance(const SpectralState &st, double theta, double gamma, double wavelength) {
    const int low_wl = (int)((wavelength - 320.0) / 40.0);
    if (low_wl < 0 || low_wl >= 11) return 0.0f;
    const double interp = std::fmod((wavelength - 320.0) / 40.0, 1.0);
    const double val_low = radiance_internal(st.configs[low_wl], theta, gamma) * st.radiances[low_wl] * 1.0;
    if (interp < 1e-6) return val_low;
    double result = (1.0 - interp) * val_low;
    if (low_wl + 1 < 11) result += interp * radiance_internal(st.configs[low_wl + 1], theta, gamma) * st.radiances[low_wl + 1] * 1.0;
    return result;
}

// sky_model.cpp:663-692: piecewise cubic in the elevation, 45 pieces on a cube-root grid
inline double sr_internal(const SkyTables &t, int turbidity, int wl, double elevation) {
    const int pieces = 45, order = 4;
    int pos = (int)(std::pow(2.0 * elevation / PI, 1.0 / 3.0) * pieces);
    if (pos > 44) pos = 44;
    const double break_x = std::pow(((double)pos / (double)pieces), 3.0) * (PI * 0.5);
    const double *coefs = t.solar[wl].data() + (order * pieces * turbidity + order * (pos + 1) - 1);
    double res = 0.0;
    const double x = elevation - break_x;
    double x_exp = 1.0;
    for (int i = 0; i < order; ++i) {
        res += x_exp * *coefs--;
        x_exp *= x;
    }
    return res * 1.0;
}

// sky_model.cpp:694-796 (the reference asserts 320 <= wavelength <= 720 and 1 <= turbidity <= 10)
inline double solar_radiance_internal2(const SkyTables &t, const SpectralState &st, double wavelength, double elevation, double gamma) {
    int turb_low = (int)st.turbidity - 1;
    double turb_frac = st.turbidity - (double)(turb_low + 1);
    if (turb_low == 9) {
        turb_low = 8;
        turb_frac = 1.0;
    }
    int wl_low = (int)((wavelength - 320.0) / 40.0);
    double wl_frac = std::fmod(wavelength, 40.0) / 40.0;
    if (wl_low == 10) {
        wl_low = 9;
        wl_frac = 1.0;
    }
    double direct_radiance = (1.0 - turb_frac) * ((1.0 - wl_frac) * sr_internal(t, turb_low, wl_low, elevation) + wl_frac * sr_internal(t, turb_low, wl_low + 1, elevation)) +
                             turb_frac * ((1.0 - wl_frac) * sr_internal(t, turb_low + 1, wl_low, elevation) + wl_frac * sr_internal(t, turb_low + 1, wl_low + 1, elevation));
    double ld[6];
    for (int i = 0; i < 6; i++) ld[i] = (1.0 - wl_frac) * t.limb[wl_low][i] + wl_frac * t.limb[wl_low + 1][i];
    const double sol_rad_sin = std::sin(st.solar_radius);
    const double ar2 = 1 / (sol_rad_sin * sol_rad_sin);
    const double singamma = std::sin(gamma);
    double sc2 = 1.0 - ar2 * singamma * singamma;
    if (sc2 < 0.0) sc2 = 0.0;
    const double sampleCosine = std::sqrt(sc2);
    const double darkeningFactor = ld[0] + ld[1] * sampleCosine + ld[2] * std::pow(sampleCosine, 2.0) + ld[3] * std::pow(sampleCosine, 3.0) +
                                   ld[4] * std::pow(sampleCosine, 4.0) + ld[5] * std::pow(sampleCosine, 5.0);
    direct_radiance *= darkeningFactor;
    return direct_radiance;
}

} // namespace skyfit

// update_sky_light (vulkan/render_sky.cpp:25-72): SkyModelParams, sun direction / disc / radiance for a scene state. light_count: the
// number of triangle lights (the sun's sampling probability is 1/2 with any, 1 without). normal_z_scale is not touched.
// Returns false (out untouched) when the tables lack the RGB datasets; without the spectral data the sun's radiance is 0.
inline bool fit_sky(const SkyTables &t, const float sun_dir_in[3], float turbidity, const float albedo[3], int light_count, RptrSceneParams &out) {
    if (!t.has_rgb()) return false;
    using namespace skyfit;
    const float l = std::sqrt(sun_dir_in[0] * sun_dir_in[0] + sun_dir_in[1] * sun_dir_in[1] + sun_dir_in[2] * sun_dir_in[2]);
    const float inv = 1.0f / l; // glm::normalize = v * inversesqrt(dot(v, v))
    const float sun_dir[3] = {sun_dir_in[0] * inv, sun_dir_in[1] * inv, sun_dir_in[2] * inv};
    const float albedo_avg = albedo[0] * 0.3333f + albedo[1] * 0.3333f + albedo[2] * 0.3333f; // dot(albedo, vec3(0.3333f))
    // arhosek_rgb_skymodelstate_alloc_init(turbidity, albedo, elevation = sun_dir.y): the float arguments widen to double
    const double T = turbidity, A = albedo_avg, E = sun_dir[1];
    double configs[3][9], radiances[3];
    for (int c = 0; c < 3; ++c) {
        cook_configuration(t.rgb[c].data(), configs[c], T, A, E);
        radiances[c] = cook_radiance_configuration(t.rgb_rad[c].data(), T, A, E);
    }
    std::memset(&out.sky_params, 0, sizeof(out.sky_params));
    for (int i = 0; i < 9; ++i)
        for (int c = 0; c < 3; ++c) out.sky_params.configs[i][c] = (float)configs[c][i];
    for (int c = 0; c < 3; ++c) out.sky_params.radiances[c] = (float)radiances[c];
    std::memcpy(out.sun_dir, sun_dir, sizeof(sun_dir));
    out.sun_cos_angle = std::cos((0.53f * 0.01745329251994329576923690768489f) / 2.0f); // cos(radians(0.53) / 2)
    for (int k = 0; k < 4; ++k) out.sun_radiance[k] = 0.0f;
    if (t.has_sun()) {
        SpectralState sun; // arhosekskymodelstate_alloc_init(state.elevation, state.turbidity, state.albedo)
        sun.solar_radius = (0.51 * (PI / 180.0)) / 2.0;
        sun.turbidity = T;
        sun.albedo = A;
        sun.elevation = E;
        for (int wl = 0; wl < 11; ++wl) {
            cook_configuration(t.spec[wl].data(), sun.configs[wl], T, A, E);
            sun.radiances[wl] = cook_radiance_configuration(t.spec_rad[wl].data(), T, A, E);
        }
        const float CIE_MIN = 360.f, CIE_MAX = 830.f;
        const int CIE_SAMPLES = 95;
        float xyz[3] = {0, 0, 0};
        int numSamples = 0;
        float last_wavelength = CIE_MIN;
        const float *TX = t.cie.data(), *TY = TX + CIE_SAMPLES, *TZ = TX + 2 * CIE_SAMPLES;
        for (int i = 0; i < CIE_SAMPLES; ++i) {
            const float wavelength = float(i) * float(CIE_MAX - CIE_MIN) / float(CIE_SAMPLES - 1) + float(CIE_MIN);
            if (wavelength > 720.0f) break; // higher wavelengths are not supported by the sky model
            // arhosekskymodel_solar_radiance(&sunState, theta = sun_dir.y, gamma = 0, wavelength) = direct + in-scattered, stored as float;
            // then the in-scattered part is taken off again (float - double, rounded to float)
            const double theta = sun_dir[1];
            const double inscattered = spectral_radiance(sun, theta, 0.0, wavelength);
            float radiance = (float)(solar_radiance_internal2(t, sun, wavelength, ((PI / 2.0) - theta), 0.0) + inscattered);
            radiance = (float)((double)radiance - inscattered);
            xyz[0] += TX[i] * radiance;
            xyz[1] += TY[i] * radiance;
            xyz[2] += TZ[i] * radiance;
            ++numSamples;
            last_wavelength = wavelength;
        }
        const float scale = float(last_wavelength - CIE_MIN) / float(numSamples);
        for (int k = 0; k < 3; ++k) xyz[k] *= scale;
        const float M[3][3] = {{3.240479f, -1.537150f, -0.498535f}, {-0.969256f, 1.875991f, 0.041556f}, {0.055648f, -0.204043f, 1.057311f}}; // xyz_to_srgb
        if (sun_dir[1] > 0.0f && xyz[0] >= 0.0f && xyz[1] >= 0.0f && xyz[2] >= 0.0f) {
            for (int r = 0; r < 3; ++r) out.sun_radiance[r] = 0.01f * ((M[r][0] * xyz[0] + M[r][1] * xyz[1]) + M[r][2] * xyz[2]);
            out.sun_radiance[3] = 1.0f;
        }
    }
    if (light_count > 0)
        out.sun_radiance[3] *= 0.5f;
    else
        out.sun_radiance[3] = 1.0f;
    return true;
}

// the "Sun" sliders of the reference's scene state (libapp/scene_state.h:79-96): height in degrees above the horizon and angle about
// the vertical -> direction; and back (what the sliders show for a direction)
inline void sun_dir_from_height_angle(float height_deg, float angle_deg, float out[3]) {
    const float rad = 0.01745329251994329576923690768489f;
    const float cosTheta = std::cos(rad * (90.0f - height_deg)), sinTheta = std::sin(rad * (90.0f - height_deg));
    out[0] = std::cos(rad * angle_deg) * sinTheta;
    out[1] = cosTheta;
    out[2] = std::sin(rad * angle_deg) * sinTheta;
}
inline void sun_height_angle_from_dir(const float d[3], float &height_deg, float &angle_deg) {
    const float deg = 57.295779513082320876798154814105f;
    height_deg = 90.0f - deg * std::acos(d[1]);
    angle_deg = deg * std::atan2(d[2], d[0]);
}

} // namespace rptr
