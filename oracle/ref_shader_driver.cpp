// ref_shader_driver.cpp -- TEST INFRASTRUCTURE, built only where a real GLM is present (`make -C oracle ref_shaders GLM_ROOT=<dir holding glm/glm.hpp>`).
//
// The reference's shader library is dual-language: the .glsl files under rendering/ compile as C++ against GLM
// (rendering/tests/compile.cpp:1-41 is the reference's own proof: it includes them in exactly this way). This driver includes the
// reference's files WHERE THEY LIE (no copies, no stand-in headers: without GLM it does not build, and the recipe says so), calls the
// functions that decide a pixel on seeded inputs and prints inputs + outputs as JSON: the golden vectors that pin the oracle
// (oracle/oshade.h) and, through it, the device code (csrc/dshade.h) for
//   * the glTF BSDF of the shipped build: sample_gltf_brdf, gltf_bsdf, gltf_wpdf (rendering/bsdfs/gltf_bsdf.glsl:294-650),
//   * binned-RIS triangle-light sampling: sample_tri_lights (rendering/mc/lights_linear.glsl:19-127) over a fixed light table,
//   * the LCG / murmur3 generator (rendering/pointsets/lcg_rng.glsl) as a cross-check of the one value SURVEY.md quotes.
// Output: tests/golden/ref_shaders.json (written by `make ref_shaders`); read by tests/test_ref_shaders.py on the CPU (oracle) and, -m gpu,
// through images. Until that file exists those tests skip with this reason.
//
// NOTE for whoever runs this first: it has never been compiled (the build container has no GLM). It follows compile.cpp's include order
// and macro hooks to the letter; if the reference's headers need one more hook, add it HERE, not in a stand-in header.
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

static const int num_lights = 40;
static TriLightData lights[num_lights + 16] = {}; // (padded with a zeroed bin: sample_tri_lights may read light_id == bin_end)
static int bin_size = 16;

#define SCENE_GET_LIGHT_SOURCE(light_id) decode_tri_light(lights[light_id])
#define SCENE_GET_LIGHT_SOURCE_COUNT() int(num_lights)
// the megakernel's definitions (vulkan/pt_megakernel.glsl:101-103) over this driver's table
#define BINNED_LIGHTS_BIN_MAX_SIZE 16
#define BINNED_LIGHTS_BIN_SIZE int(bin_size)
#define SCENE_GET_BINNED_LIGHTS_BIN_COUNT() ((int(num_lights) + (bin_size - 1)) / int(bin_size))

namespace binned {
#include "rendering/mc/lights_linear.glsl"
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
    // ---- RNG (SURVEY 8a2's known answer: index 3, frame offset 7, pixel (10, 20) of a 256 x 256 frame -> the state AFTER the first draw is
    // 1349923967 and the draw 0.314303666; tests/test_oracle.py holds the oracle to it, tests/test_ref_shaders.py to what this prints)
    {
        using namespace ref_shaders;
        LCGRand rng = get_lcg_rng(3u, 7u, glm::uvec4(10u, 20u, 256u, 256u));
        const uint32_t s0 = rng.state;
        const float f0 = lcg_randomf(rng);
        std::printf("\"rng\": {\"state\": %u, \"first\": %.9g},\n", s0, f0);
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
    for (int k = 0; k < num_lights; ++k) {
        const glm::vec3 c(6.0f * U(gen) - 3.0f, 2.0f + U(gen), 6.0f * U(gen) - 3.0f);
        const glm::vec3 a = c + 0.3f * unit(), b = c + 0.3f * unit(), d = c + 0.3f * unit();
        TriLightData &t = lights[k];
        t.v0_x = a.x; t.v0_y = a.y; t.v0_z = a.z; t.v1_x = b.x; t.v1_y = b.y; t.v1_z = b.z; t.v2_x = d.x; t.v2_y = d.y; t.v2_z = d.z;
        t.radiance_x = 1.0f + 9.0f * U(gen); t.radiance_y = 1.0f + 9.0f * U(gen); t.radiance_z = 1.0f + 9.0f * U(gen);
    }
    std::printf("\"lights\": [");
    for (int k = 0; k < num_lights; ++k)
        std::printf("[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]%s", lights[k].v0_x, lights[k].v0_y, lights[k].v0_z, lights[k].v1_x,
                    lights[k].v1_y, lights[k].v1_z, lights[k].v2_x, lights[k].v2_y, lights[k].v2_z, lights[k].radiance_x, lights[k].radiance_y, lights[k].radiance_z,
                    k + 1 < num_lights ? ", " : "");
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
    std::printf("]\n}\n");
    return 0;
}
