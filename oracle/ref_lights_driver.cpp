// ref_lights_driver.cpp -- TEST INFRASTRUCTURE, built only where a real GLM is present (`make -C oracle ref_shaders GLM_ROOT=<dir holding glm/glm.hpp>`).
//
// SURVEY.md section 8(c) group (11): the host-side preparation of the emitter table that next-event estimation reads -- librender/lights.cpp compiled
// UNMODIFIED where it lies (the #include below is the whole of it): update_light_sampling = estimate_normalized_radiance -> trim_dim_emitters ->
// equalize_emitter_bins (lights.cpp:75-90,166-349; the Halton shuffle, the clone / split rounds, the final bin order). lights.cpp needs GLM and
// nothing else outside the reference's librender/ + util/ headers; the functions of it that read a Scene (collect_emitters) are compiled but never
// called, so the program links with unresolved symbols ignored (the recipe says how).
// The product's own preparation (realtimepathtracingresearchframework_amd/lights.py = host/lights.hpp, byte-identical to each other by
// tests/test_validation_cli.py) is held to this program's output by tests/test_ref_shaders.py: same emitters in the same order, radiances to 1e-6
// (one atan per emitter goes through the host's libm).
// Output: the file named on the command line (the recipe: tests/golden/ref_lights.json). Until the file exists the test skips.
//
// NOTE for whoever runs this first: never compiled against a real GLM (the build container has none). <numeric> is included first because
// lights.cpp uses std::partial_sum without including it.
#include <numeric>

#include "librender/lights.cpp"

#include <cstdio>
#include <random>

struct Case {
    const char *name;
    int emitters;
    int bin_size;
    float min_radiance;
    float min_perceived_receiver_dist;
    float size_lo, size_hi;     // edge length range
    float rad_lo, rad_hi;       // radiance range (log-uniform)
    int degenerate_every;       // > 0: every n-th emitter has zero area
};

int main(int argc, char **argv) {
    // (lights.cpp reports its re-binning on stdout: the vectors go to the file named on the command line)
    if (argc < 2) {
        std::fprintf(stderr, "usage: ref_lights <output.json>\n");
        return 2;
    }
    FILE *out = std::fopen(argv[1], "w");
    if (!out) return 2;
    const Case cases[] = {
        {"c3_like_256_equal_quads", 512, 16, 0.0f, 15.0f, 0.5f, 0.5f, 20.0f, 20.0f, 0},
        {"forty_mixed", 40, 16, 0.0f, 15.0f, 0.1f, 2.0f, 0.5f, 200.0f, 0},
        {"five_emitters_one_bin", 5, 16, 0.0f, 15.0f, 0.2f, 1.0f, 1.0f, 50.0f, 0},
        {"seventeen_bin_of_eight", 17, 8, 0.0f, 15.0f, 0.1f, 3.0f, 0.1f, 1000.0f, 0},
        {"trimmed_and_degenerate", 96, 16, 0.02f, 15.0f, 0.05f, 1.5f, 0.2f, 400.0f, 7},
        {"near_receiver", 64, 4, 0.0f, 1.0f, 0.05f, 4.0f, 1.0f, 30.0f, 0},
        {"bin_size_one_is_left_alone", 12, 1, 0.0f, 15.0f, 0.3f, 0.6f, 1.0f, 9.0f, 0},
    };
    std::mt19937 gen(20240923u);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    auto rnd3 = [&](float scale, float offset) { // (separate statements: the printed inputs must not depend on the compiler's argument order)
        glm::vec3 v;
        v.x = scale * U(gen) + offset;
        v.y = scale * U(gen) + offset;
        v.z = scale * U(gen) + offset;
        return v;
    };
    auto print_lights = [out](const char *key, const std::vector<TriLight> &ls) {
        std::fprintf(out, "\"%s\": [", key);
        for (size_t i = 0; i < ls.size(); ++i) {
            const TriLight &l = ls[i];
            std::fprintf(out, "%s[%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g]", i ? ", " : "", l.v0.x, l.v0.y, l.v0.z, l.v1.x, l.v1.y, l.v1.z, l.v2.x,
                        l.v2.y, l.v2.z, l.radiance.x, l.radiance.y, l.radiance.z);
        }
        std::fprintf(out, "]");
    };
    std::fprintf(out, "{\"cases\": [\n");
    const int n_cases = int(sizeof(cases) / sizeof(cases[0]));
    for (int k = 0; k < n_cases; ++k) {
        const Case &c = cases[k];
        std::vector<TriLight> emitters(c.emitters);
        for (int i = 0; i < c.emitters; ++i) {
            TriLight &l = emitters[i];
            const glm::vec3 centre = rnd3(40.0f, -20.0f);
            const float size = c.size_lo + (c.size_hi - c.size_lo) * U(gen);
            const glm::vec3 e1 = rnd3(2.0f, -1.0f);
            const glm::vec3 e2 = rnd3(2.0f, -1.0f);
            l.v0 = centre;
            l.v1 = centre + size * e1;
            l.v2 = (c.degenerate_every > 0 && i % c.degenerate_every == 3) ? centre + (2.0f * size) * e1 : centre + size * e2;
            const float lum = c.rad_lo * std::pow(c.rad_hi / c.rad_lo, U(gen));
            const glm::vec3 tint = rnd3(0.5f, 0.5f);
            l.radiance = lum * tint;
        }
        LightSamplingConfig params;
        params.bin_size = c.bin_size;
        params.min_radiance = c.min_radiance;
        params.min_perceived_receiver_dist = c.min_perceived_receiver_dist;
        BinnedLightSampling binned;
        update_light_sampling(binned, emitters, params);
        std::fprintf(out, " {\"name\": \"%s\", \"bin_size\": %d, \"min_radiance\": %.9g, \"min_perceived_receiver_dist\": %.9g,\n  ", c.name, c.bin_size, c.min_radiance,
                    c.min_perceived_receiver_dist);
        print_lights("emitters", emitters);
        std::fprintf(out, ",\n  ");
        print_lights("binned_emitters", binned.emitters);
        std::fprintf(out, ",\n  \"binned_radiances\": [");
        for (size_t i = 0; i < binned.radiances.size(); ++i) std::fprintf(out, "%s%.9g", i ? ", " : "", binned.radiances[i]);
        std::fprintf(out, "]}%s\n", k + 1 < n_cases ? "," : "");
    }
    std::fprintf(out, "]}\n");
    std::fclose(out);
    return 0;
}
