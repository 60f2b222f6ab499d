// rptr_validate.cpp -- headless validation run through the C ABI, the HIP backend's counterpart of
//   rptr <scene> --backend ... --validation <prefix> --validation-spp <n> --img <w> <h> --pfm
// (reference: cmdline.cpp:42-50,377-386; frame loop app.cpp:350-522; handle_mode_actions
// libapp/app_state.cpp:464-481; WriteImage::write_pfm util/write_image.cpp:34-64).
// One frame = params.batch_spp samples (1 by default, like the reference); frames accumulate until the target; the
// float accumulation buffer is read back through RenderBackend::readback_framebuffer(float*) and written as
// <prefix>_%04d.pfm (bottom-up RGB float32, "PF\n<w> <h>\n-1.0\n"). The scene comes from a dump file
// (scene_dump.hpp) instead of a .vks.
//
//   rptr_validate <scene.rpsc> --validation <prefix> [--validation-spp n] [--img w h] [--variant gltf|diffuse]
//                 [--batch-spp k] [--every-frame]
#include "render_hip.hpp"
#include "scene_dump.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static bool write_pfm(const std::string &prefix, unsigned width, unsigned height, unsigned channels, const float *pixels) {
    if (width == 0 || height == 0 || channels < 3 || !pixels) return false;
    const std::string path = prefix + ".pfm";
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "PF\n%i %i\n-1.0\n", width, height);
    std::vector<float> rgb((size_t)width * height * 3);
    for (unsigned y = 0; y < height; ++y) // the file stores the bottom row first
        for (unsigned x = 0; x < width; ++x)
            for (unsigned j = 0; j < 3; ++j) rgb[((size_t)width * (height - y - 1) + x) * 3 + j] = pixels[((size_t)width * y + x) * channels + j];
    const bool ok = std::fwrite(rgb.data(), sizeof(float), rgb.size(), f) == rgb.size();
    std::fclose(f);
    return ok;
}

int main(int argc, char **argv) {
    std::string scene_path, prefix;
    int target_spp = 1, width = 256, height = 256, variant = RPTR_VARIANT_GLTF, batch_spp = 1;
    bool every_frame = false, describe = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int k) {
            if (i + k >= argc) {
                std::fprintf(stderr, "%s needs %d argument(s)\n", a.c_str(), k);
                std::exit(2);
            }
        };
        if (a == "--validation") { need(1); prefix = argv[++i]; }
        else if (a == "--validation-spp") { need(1); target_spp = std::atoi(argv[++i]); }
        else if (a == "--img") { need(2); width = std::atoi(argv[++i]); height = std::atoi(argv[++i]); }
        else if (a == "--batch-spp") { need(1); batch_spp = std::atoi(argv[++i]); }
        else if (a == "--variant") { need(1); variant = std::strcmp(argv[++i], "diffuse") == 0 ? RPTR_VARIANT_SIMPLE : RPTR_VARIANT_GLTF; }
        else if (a == "--every-frame") every_frame = true;
        else if (a == "--describe") describe = true; // load the scene, print what was read, do not render
        else if (a == "--pfm") {} // the only format of this tool
        else if (a[0] != '-') scene_path = a;
        else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
    }
    if (describe && !scene_path.empty()) {
        try {
            const rptr::SceneDump s = rptr::SceneDump::load(scene_path);
            unsigned long long tris = 0, qsum = 0;
            for (size_t i = 0; i < s.geometries.size(); ++i) {
                tris += s.geometries[i].num_tris;
                for (uint64_t q : s.qpos[i]) qsum += q & 0xFFFFFFull;
            }
            std::printf("geometries %zu meshes %zu parameterized_meshes %zu instances %zu materials %zu lights %zu triangles %llu qsum %llu fovy %.6f "
                        "max_path_depth %d bin_size %d sun_w %.6f\n",
                        s.geometries.size(), s.meshes.size(), s.pmeshes.size(), s.instances.size(), s.materials.size(), s.lights.size(), tris, qsum,
                        s.camera.fovy, s.render_params.max_path_depth, s.lighting.bin_size, s.scene_params.sun_radiance[3]);
            return 0;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "rptr_validate: %s\n", e.what());
            return 3;
        }
    }
    if (scene_path.empty() || prefix.empty() || target_spp < 1 || batch_spp < 1 || width < 1 || height < 1) {
        std::fprintf(stderr, "usage: rptr_validate <scene.rpsc> --validation <prefix> [--validation-spp n] [--img w h] [--variant gltf|diffuse] "
                             "[--batch-spp k] [--every-frame]\n");
        return 2;
    }
    try {
        rptr::SceneDump scene = rptr::SceneDump::load(scene_path);
        rptr::RenderHip backend;
        backend.initialize(width, height);
        backend.set_scene(scene.desc());
        backend.params = scene.render_params;
        backend.params.batch_spp = batch_spp;
        backend.lighting_params = scene.lighting;
        backend.update_config(scene.scene_params);
        rptr::RenderConfiguration cfg{};
        std::memcpy(cfg.camera.pos, scene.camera.pos, 12);
        std::memcpy(cfg.camera.dir, scene.camera.dir, 12);
        std::memcpy(cfg.camera.up, scene.camera.up, 12);
        cfg.camera.fovy = scene.camera.fovy;
        cfg.active_variant = variant;
        cfg.reset_accumulation = true; // frame 0 of the accumulation (app.cpp: reset on scene load)
        std::vector<float> img((size_t)width * height * 4);
        int accumulated = 0;
        double gpu_ms = 0.0;
        while (accumulated < target_spp) {
            const rptr::RenderStats st = backend.render(cfg); // params.batch_spp samples
            cfg.reset_accumulation = false;
            accumulated = st.spp;
            gpu_ms += st.render_time;
            const bool done = accumulated >= target_spp;
            if (done || every_frame) {
                if (backend.readback_framebuffer(img.size(), img.data()) != img.size()) throw std::runtime_error("read-back failed");
                char name[32];
                std::snprintf(name, sizeof(name), "_%04d", accumulated);
                if (!write_pfm(prefix + name, (unsigned)width, (unsigned)height, 4, img.data())) throw std::runtime_error("cannot write " + prefix + name + ".pfm");
            }
        }
        std::printf("%s: %d spp in %.3f ms GPU time -> %s_%04d.pfm\n", backend.name().c_str(), accumulated, gpu_ms, prefix.c_str(), accumulated);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rptr_validate: %s\n", e.what());
        return 3;
    }
    return 0;
}
