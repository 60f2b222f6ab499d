// rptr_cli.cpp -- the reference's headless run modes through the C ABI (binary: bin/rptr_hip):
//
//   <scene> = the reference's .vks file (textures in <name>_textures/) or the flat dump scenes.py writes (.rpsc)
//   rptr_hip <scene> --validation <prefix> [--validation-spp n] [--img w h] [--pfm]
//   rptr_hip <scene.rpsc> --profiling <csv prefix> [--profiling-fps f] [--profiling-img <prefix>] [--profiling-frames n]
//            [--animate-wave amplitude kx] [--synchronous] [--fly-through] [--frames-in-flight n [--frames-per-launch b]]
//   common:  [--eye x y z] [--center x y z] [--up x y z] [--fov deg] [--variant gltf|diffuse|gltf-transmission] [--batch-spp k] [--every-frame]
//            [--config file.ini]... [--keyframe [<seconds>:]file.ini]... [--camera n] [--freeze-frame] [--upscale n] [--backend hip]
//            [--sky-data <dir with the Hosek-Wilkie data headers>]   (Sun settings of the .ini files refit the sky: host/sky_fit.hpp; also RPTR_SKY_DATA)
//            [--rng-variant uniform|bn|sobol|z-sobol] [--bn-table bn_tables.h|BNData.u32] [--force-bvh-rebuild] [--rebuild-triangle-budget n]
//            [--devices n | --devices a,b,c] [--stripe-rows r]
//
// --config / --keyframe read the reference's .ini files (ini_config.hpp); in profiling mode every keyframe is held for its length
// (default 1 s of animation time = --profiling-fps frames) and the accumulation restarts when the keyframe changes.
// --devices: one process, several GPUs (render_group.hpp): the frame is cut into stripes of --stripe-rows rows, stripe s -> device
// s % n, tile radiance gathered to the first device with the library's RCCL gather (peer copies when a device is listed twice).
//
// Flag names, file names and the CSV header are the reference's (cmdline.cpp:10-104,296-474; libapp/app_state.cpp:218-255,
// 291-322,464-498; libapp/benchmark_info.cpp:69-124; util/write_image.cpp:34-64):
//  * validation mode renders at time 0, one frame = params.batch_spp samples, accumulates until --validation-spp and writes
//    the float accumulation buffer as <prefix>_%04d.pfm (accumulated spp in the name); n < 1 = after every frame;
//  * profiling mode advances time by 1/fps per frame (non-realtime), appends one CSV row per frame to <prefix>.csv with the
//    header frames_total,keyframe,frames_accumulated,render_time_ms,app_time_ms, and writes <img prefix>_%04d.pfm once per
//    second of animation time. The run ends at its last keyframe (--keyframe [len:]file.ini ...); without keyframes the reference
//    stops after its first frame (imstate.cpp:890-898) and so does this host unless --profiling-count n (its own flag) asks for n
//    frames, a keyframe then being one second. --profiling-frames is the reference's old spelling of --profiling-fps.
//    The loop is the reference's (app.cpp:453-469): begin_frame / draw_frame / end_frame with a command stream -- two frames in flight,
//    render_time_ms of a row = the statistics the backend has at that point, i.e. of the frame two submissions earlier
//    (RenderStats::frame_stats_delay, vulkan/render_vulkan.cpp:2229-2243) -- unless --synchronous (the application's "force synchronous
//    rendering": cmd_stream = nullptr) is given. --frames-in-flight n [--frames-per-launch b] (this host's) queues deeper than the reference's
//    two swap buffers: launch sequences of b frames, each frame with its own camera, n sequences in flight -- bench.py's schedule;
//    keyframes and --fly-through (this host's: the camera path of bench.py, yaw 0.002 rad and 2 cm sideways per frame, 64 views around the scene's own, a
//    moved camera restarts the accumulation) work in every schedule.
//  * --animate-wave a k (profiling mode, scenes whose mesh 0 is dynamic): y = y0 + a sin(k x + 2 pi t) on geometry 0 before
//    every frame, followed by rptr_hip_refit -- SURVEY 8d C5 (the reference animates with a compute shader,
//    render_vulkan.cpp:2834-2840; per-frame BLAS update + TLAS refit :1323-1354).
//  * data-capture mode (--data-capture <prefix> [--data-capture-spp n] [--data-capture-no-rgba] [--data-capture-no-aovs]
//    [--data-capture-albedo-roughness|-normal-depth|-motion]) stores the accumulation buffer and the chosen AOV images of every
//    keyframe as <prefix>_%04d_{rgba,albedo_roughness,normal_depth,motion_jitter}.exr (libapp/app_state.cpp:499-531).
// Images: --exr (default, as in the reference), --pfm, --png (write_image.hpp).
// The scene comes from a dump file (scene_dump.hpp) instead of a .vks (python -m ...vks converts).
#include "ini_config.hpp"
#include "render_group.hpp"
#include "scene_dump.hpp"
#include "sky_fit.hpp"
#include "vks_reader.hpp"
#include "write_image.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

enum OutputFormat { FORMAT_EXR, FORMAT_PFM, FORMAT_PNG }; // cmdline.cpp:450-460: EXR is the default

// BasicApplicationState::save_framebuffer (libapp/app_state.cpp:341-438): the float accumulation buffer as EXR / PFM, the 8-bit
// frame buffer as PNG, under <prefix>_<number> (+ suffix)
static void save_image(rptr::RenderGroup &backend, OutputFormat format, const std::string &prefix, int number, const std::string &suffix, int width, int height,
                       std::vector<float> &img) {
    char name[32];
    std::snprintf(name, sizeof(name), "_%04d", number);
    const std::string base = prefix + name + suffix;
    bool ok = false;
    if (format == FORMAT_PNG) {
        std::vector<unsigned char> rgba8((size_t)width * height * 4);
        if (backend.readback_framebuffer(rgba8.size(), rgba8.data()) != rgba8.size()) throw std::runtime_error("read-back failed");
        ok = rptr::write_png(base, (unsigned)width, (unsigned)height, 4, rgba8.data());
    } else {
        if (backend.readback_framebuffer(img.size(), img.data()) != img.size()) throw std::runtime_error("read-back failed");
        ok = format == FORMAT_PFM ? rptr::write_pfm(base, (unsigned)width, (unsigned)height, 4, img.data())
                                  : rptr::write_exr<float>(base, (unsigned)width, (unsigned)height, 4, img.data());
    }
    if (!ok) throw std::runtime_error("cannot write " + base);
}
// BasicApplicationState::save_aov_exr (libapp/app_state.cpp:441-462): an AOV image as a HALF EXR
static void save_aov(rptr::RenderGroup &backend, rptr::RenderHip::AOVBufferIndex aov, const std::string &base, int width, int height) {
    std::vector<uint16_t> half((size_t)width * height * 4);
    if (backend.readback_aov(aov, half.size(), half.data()) != half.size()) throw std::runtime_error("AOV read-back failed");
    if (!rptr::write_exr<uint16_t>(base, (unsigned)width, (unsigned)height, 4, half.data())) throw std::runtime_error("cannot write " + base + ".exr");
}

// package data (data/) sits next to librptr_hip.so: the directory of the library this program is linked against
static std::string data_dir() {
    if (const char *e = std::getenv("RPTR_DATA_DIR")) return e;
    Dl_info info;
    if (dladdr((const void *)&rptr_hip_name, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        const size_t k = p.rfind('/');
        return (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/data";
    }
    return "data";
}

// a scene file: the reference's own .vks (with its _textures directory), or the flat dump of scenes.py
static rptr::vks::LoadParams g_load; // --remove-first-lods, --instance-pruning, --small-deformation, --ignore-animation, --ignore-textures, --load-specularity (SceneLoaderParams::PerFile)
static rptr::SceneDump load_one_scene(const std::string &path) {
    const size_t n = path.size();
    if (n >= 4 && path.compare(n - 4, 4, ".vks") == 0) return rptr::vks::read_scene(path, data_dir(), g_load);
    return rptr::SceneDump::load(path);
}
// `<scene_file> [<scene_file>...]` (Scene::Scene(fnames, ...), librender/scene.cpp:50-69): every further file is appended to the first --
// meshes, parameterized meshes, instances, materials, textures, indices shifted; camera and parameters are the first file's -- then
// `--deduplicate-scene` merges equal meshes / materials / textures and drops what nothing refers to (Scene::deduplicate + garbage_collect),
// and the emitters are collected and binned again over the whole scene (librender/lights.cpp through host/lights.hpp).
static std::vector<std::string> g_more_scenes;
static bool g_deduplicate = false, g_merge_partitions = false; // --deduplicate-scene, --merge-partition-instances (SceneLoaderParams)
static rptr::SceneDump load_scene(const std::string &path) {
    auto one = [&](const std::string &file) {
        rptr::SceneDump d = load_one_scene(file);
        if (g_merge_partitions) {
            const size_t n = d.merge_partition_instances();
            if (n) std::printf("%s: merged %zu partition instances\n", file.c_str(), n);
        }
        return d;
    };
    rptr::SceneDump s = one(path);
    for (const std::string &more : g_more_scenes) s.append(one(more));
    if (g_deduplicate) {
        const rptr::SceneDump::DedupStats st = s.deduplicate();
        if (st.meshes) std::printf("Duplicate geometry detected! Removed %zu meshes\n", st.meshes);
        if (st.materials) std::printf("Removed %zu unused materials\n", st.materials);
        if (st.textures) std::printf("Removed %zu unused textures\n", st.textures);
    }
    if (!g_more_scenes.empty() || g_deduplicate || g_merge_partitions) rptr::lights::prepare_lights(s);
    return s;
}

int main(int argc, char **argv) {
    std::string scene_path, validation_prefix, csv_prefix, profiling_img_prefix, capture_prefix;
    OutputFormat format = FORMAT_EXR;
    bool data_capture = false, capture_rgba = true, capture_albedo = true, capture_normal = true, capture_motion = true; // DataCaptureConfig, libapp/shell.h
    bool have_profiling_options = false, want_help = false;
    float capture_fps = 60.f;
    int capture_spp = 1;
    int target_spp = 1, width = 256, height = 256, variant = RPTR_VARIANT_GLTF, batch_spp = 1, profiling_frames = 1; // without keyframes the reference's profiling run ends after its first frame (imstate.cpp:890-898)
    float profiling_fps = 60.f, wave_amp = 0.f, wave_k = 0.f;
    float eye[3], center[3], up[3] = {0, 1, 0}, fov = 0.f;
    bool every_frame = false, describe = false, validation = false, profiling = false, got_eye = false, got_center = false, got_up = false;
    bool freeze_frame = false, got_batch_spp = false, got_variant = false;
    int rng_variant = -1, force_bvh_rebuild = -1, rebuild_triangle_budget = -1; // -1: as the configuration files say
    std::string bn_table_path, dump_scene_path, sky_data;
    bool got_frames_per_launch = false;
    int upscale = 0, stripe_rows = 8, frames_in_flight = 0, frames_per_launch = 1; // frames_in_flight 0: the reference's loop (two swap buffers)
    bool synchronous = false, fly_through = false;
    std::vector<int> devices{0};
    std::vector<std::string> config_inis;
    struct Keyframe {
        std::string ini;
        double hold;
    };
    std::vector<Keyframe> keyframes;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int k) {
            if (i + k >= argc) {
                std::fprintf(stderr, "%s needs %d argument(s)\n", a.c_str(), k);
                std::exit(2);
            }
        };
        auto vec3 = [&](float *v) {
            need(3);
            for (int k = 0; k < 3; ++k) v[k] = (float)std::atof(argv[++i]);
        };
        if (a == "--validation") { need(1); validation_prefix = argv[++i]; validation = true; }
        else if (a == "--validation-spp") { need(1); target_spp = std::atoi(argv[++i]); }
        else if (a == "--profiling") { need(1); csv_prefix = argv[++i]; profiling = true; }
        else if (a == "--profiling-fps") { need(1); profiling_fps = (float)std::atof(argv[++i]); if (profiling_fps <= 1) profiling_fps = 1; have_profiling_options = true; }
        else if (a == "--profiling-img") { need(1); profiling_img_prefix = argv[++i]; have_profiling_options = true; }
        else if (a == "--benchmark-file") { std::fprintf(stderr, "--benchmark-file <name>.csv is now --profiling <name>\n"); return 2; } // cmdline.cpp:409-413
        else if (a == "--profiling-frames") { // the reference's old spelling of --profiling-fps (cmdline.cpp:397-403)
            need(1); profiling_fps = (float)std::atof(argv[++i]); if (profiling_fps <= 1) profiling_fps = 1; have_profiling_options = true; }
        else if (a == "--frames-in-flight") { need(1); frames_in_flight = std::max(1, std::min(16, std::atoi(argv[++i]))); }   // (this host's: the pipelined schedule of bench.py)
        else if (a == "--frames-per-launch") { need(1); frames_per_launch = std::max(1, std::min(8, std::atoi(argv[++i]))); got_frames_per_launch = true; }
        else if (a == "--synchronous") synchronous = true;   // AppState::synchronous_rendering ("force synchronous rendering", libapp/app_state.cpp:145)
        else if (a == "--fly-through") fly_through = true;   // (this host's: bench.py's camera path)
        else if (a == "--profiling-count") { need(1); profiling_frames = std::atoi(argv[++i]); have_profiling_options = true; } // (this host's: frames of a run without keyframes)
        else if (a == "--animate-wave") { need(2); wave_amp = (float)std::atof(argv[++i]); wave_k = (float)std::atof(argv[++i]); }
        else if (a == "--img") { need(2); width = std::atoi(argv[++i]); height = std::atoi(argv[++i]); }
        else if (a == "--eye") { vec3(eye); got_eye = true; }
        else if (a == "--center") { vec3(center); got_center = true; }
        else if (a == "--up") { vec3(up); got_up = true; }
        else if (a == "--fov") { need(1); fov = (float)std::atof(argv[++i]); }
        else if (a == "--batch-spp") { need(1); batch_spp = std::atoi(argv[++i]); got_batch_spp = true; }
        else if (a == "--config") { need(1); config_inis.push_back(argv[++i]); }
        else if (a == "--keyframe" || a == "--frame") { // [<length>:]<file> (cmdline.cpp:319-334; --frame: the old spelling)
            need(1);
            std::string v = argv[++i];
            double hold = 1.0;
            const size_t colon = v.find(':');
            if (colon != std::string::npos && colon > 0) {
                char *end = nullptr;
                const double len = std::strtod(v.c_str(), &end);
                if (end == v.c_str() + colon) {
                    hold = len;
                    v.erase(0, colon + 1);
                }
            }
            keyframes.push_back({v, hold});
        }
        else if (a == "--camera") { need(1); ++i; } // an index beyond the scene's cameras leaves the default view (libapp/scene_state.cpp:45-50); the scene file holds one
        else if (a == "--upscale") { need(1); upscale = std::max(1, std::atoi(argv[++i])); }
        else if (a == "--sky-data") { need(1); sky_data = argv[++i]; }
        else if (a == "--stripe-rows") { need(1); stripe_rows = std::atoi(argv[++i]); }
        else if (a == "--devices") { // a count (devices 0..n-1) or a comma-separated list of HIP ordinals
            need(1);
            const std::string v = argv[++i];
            devices.clear();
            if (v.find(',') == std::string::npos) {
                for (int d = 0; d < std::max(1, std::atoi(v.c_str())); ++d) devices.push_back(d);
            } else {
                size_t at = 0;
                while (at <= v.size()) {
                    const size_t comma = v.find(',', at);
                    devices.push_back(std::atoi(v.substr(at, comma == std::string::npos ? std::string::npos : comma - at).c_str()));
                    if (comma == std::string::npos) break;
                    at = comma + 1;
                }
            }
        }
        else if (a == "--variant") { need(1); const char *v = argv[++i]; variant = std::strcmp(v, "diffuse") == 0 ? RPTR_VARIANT_SIMPLE : std::strcmp(v, "gltf-transmission") == 0 ? RPTR_VARIANT_GLTF_TRANSMISSION : RPTR_VARIANT_GLTF; got_variant = true; }
        else if (a == "--every-frame") every_frame = true;
        else if (a == "--remove-first-lods") { need(1); g_load.remove_first_lods = std::max(0, std::atoi(argv[++i])); }
        else if (a == "--instance-pruning") { need(1); g_load.instance_pruning_probability = (float)std::atof(argv[++i]); }
        else if (a == "--small-deformation") g_load.small_deformation = true;
        else if (a == "--ignore-animation") g_load.ignore_animation = true;
        else if (a == "--ignore-textures") g_load.ignore_textures = true;
        else if (a == "--load-specularity") g_load.load_specularity = true;
        else if (a == "--dump-scene") { need(1); dump_scene_path = argv[++i]; describe = true; } // --describe + the scene as read, in the flat layout
        else if (a == "--describe") describe = true; // load the scene, print what was read, do not render
        else if (a == "--pfm") format = FORMAT_PFM;
        else if (a == "--exr") format = FORMAT_EXR;
        else if (a == "--png") format = FORMAT_PNG;
        else if (a == "--data-capture") { need(1); capture_prefix = argv[++i]; data_capture = true; }
        else if (a == "--data-capture-spp") { need(1); capture_spp = std::max(1, std::atoi(argv[++i])); }
        else if (a == "--data-capture-fps") { need(1); capture_fps = std::max(1.0f, (float)std::atof(argv[++i])); }
        else if (a == "--data-capture-no-rgba") capture_rgba = false; // cmdline.cpp:432-452
        else if (a == "--data-capture-no-aovs") capture_albedo = capture_normal = capture_motion = false;
        else if (a == "--data-capture-albedo-roughness") capture_albedo = true;
        else if (a == "--data-capture-normal-depth") capture_normal = true;
        else if (a == "--data-capture-motion") capture_motion = true;
        else if (a == "--vulkan-device") { need(1); devices.assign(1, std::atoi(argv[++i])); } // ProgramArgs::device_override: here the HIP ordinal
        else if (a == "--resource-dir") { need(1); ++i; }      // (shader / resource search path of the reference's backends: nothing to find here)
        else if (a == "--merge-partition-instances") g_merge_partitions = true; // SceneLoaderParams::PerFile::merge_partition_instances, librender/scene.cpp:757-797
        else if (a == "--deduplicate-scene") g_deduplicate = true; // Scene::deduplicate + garbage_collect: less memory, same image
        else if (a == "-h" || a == "--help") want_help = true;
        else if (a == "--backend") { // cmdline.cpp:363-376: the last one wins; this binary hosts one
            need(1);
            const std::string b = argv[++i];
            if (b != "hip" && b != "rptr_hip") { std::fprintf(stderr, "unknown backend %s (available: hip)\n", b.c_str()); return 2; }
        }
        else if (a == "--freeze-frame") freeze_frame = true;
        else if (a == "--rng-variant") { // RenderBackendOptions::rng_variant (render_params.glsl.h:34-43,76)
            need(1);
            rng_variant = rptr::rng_variant_from_name(argv[++i]);
            if (rng_variant < 0) { std::fprintf(stderr, "unknown point set %s (uniform, bn, sobol, z-sobol)\n", argv[i]); return 2; }
        }
        else if (a == "--bn-table") { need(1); bn_table_path = argv[++i]; }
        else if (a == "--force-bvh-rebuild") force_bvh_rebuild = 1;
        else if (a == "--rebuild-triangle-budget") { need(1); rebuild_triangle_budget = std::atoi(argv[++i]); }
        else if (a == "--disable-ui") {}
        else if (a[0] != '-') {
            if (scene_path.empty()) scene_path = a;
            else g_more_scenes.push_back(a);
        }
        else {
            // cmdline.cpp:226-259: the single-dash arguments of old versions get a pointer to their successors
            static const char *old_backends[] = {"-vulkan", "-embree", "-dxr", "-optix", "-metal"};
            static const char *old_args[] = {"-img", "-config", "-validation", "-eye", "-center", "-up", "-fov", "-camera", "-spp", "-profiling-frames"};
            for (const char *o : old_backends)
                if (a == o) std::fprintf(stderr, "%s used to be a command line argument that selects a rendering backend: use --backend <BACKEND> (optional)\n", o);
            for (const char *o : old_args)
                if (a == o) std::fprintf(stderr, "%s used to be a command line argument: long-form arguments take double dashes now (-%s)\n", o, o);
            std::fprintf(stderr, "Unknown argument: %s\n", a.c_str());
            return 2;
        }
    }
    if (describe && !scene_path.empty()) {
        try {
            const rptr::SceneDump s = load_scene(scene_path);
            if (!dump_scene_path.empty()) s.save(dump_scene_path);
            unsigned long long tris = 0, qsum = 0;
            for (size_t i = 0; i < s.geometries.size(); ++i) {
                tris += s.geometries[i].num_tris;
                for (uint64_t q : s.qpos[i]) qsum += q & 0xFFFFFFull;
            }
            std::printf("geometries %zu meshes %zu parameterized_meshes %zu instances %zu materials %zu lights %zu triangles %llu qsum %llu fovy %.6f "
                        "max_path_depth %d bin_size %d sun_w %.6f\n",
                        s.geometries.size(), s.meshes.size(), s.pmeshes.size(), s.instances.size(), s.materials.size(), s.lights.size(), tris, qsum,
                        s.camera.fovy, s.render_params.max_path_depth, s.lighting.bin_size, s.scene_params.sun_radiance[3]);
            if (!config_inis.empty() || !keyframes.empty()) { // ... and what the configuration files make of it
                rptr::HostConfig c;
                c.params = s.render_params;
                c.lighting = s.lighting;
                c.camera = s.camera;
                for (const std::string &ini : config_inis) rptr::load_config(ini, c);
                std::printf("config target_spp %d batch_spp %d max_path_depth %d rr_path_depth %d glossy_only %d exposure %.6f tonemap %d output_channel %d "
                            "output_moment %d bin_size %d variant %d rng_variant %d force_bvh_rebuild %d rebuild_triangle_budget %d bump_scale %.6f sun_changed %d reprojection_mode %d raster_taa %d upscale %d "
                            "cam_pos %.6f %.6f %.6f cam_dir %.6f %.6f %.6f\n",
                            c.target_spp, c.params.batch_spp, c.params.max_path_depth, c.params.rr_path_depth, c.params.glossy_only_mode, c.params.exposure,
                            c.params.early_tone_mapping_mode, c.params.output_channel, c.params.output_moment, c.lighting.bin_size, c.variant, c.rng_variant, c.force_bvh_rebuild,
                            c.rebuild_triangle_budget, c.bump_scale, c.sun_changed ? 1 : 0, c.params.reprojection_mode, c.params.enable_raster_taa, c.params.render_upscale_factor, c.camera.pos[0], c.camera.pos[1], c.camera.pos[2], c.camera.dir[0],
                            c.camera.dir[1], c.camera.dir[2]);
                for (const Keyframe &k : keyframes) {
                    rptr::HostConfig kc = c;
                    const int blocks = rptr::load_config(k.ini, kc);
                    std::printf("keyframe hold %.6f blocks %d exposure %.6f cam_pos %.6f %.6f %.6f\n", k.hold, blocks, kc.params.exposure, kc.camera.pos[0],
                                kc.camera.pos[1], kc.camera.pos[2]);
                }
            }
            return 0;
        } catch (const std::exception &e) {
            std::fprintf(stderr, "rptr_hip: %s\n", e.what());
            return 3;
        }
    }
    if (validation && target_spp < 1) { // "< 1: write after every frame" -- needs an end here: one frame
        every_frame = true;
        target_spp = 1;
    }
    if (have_profiling_options && !profiling) { // cmdline.cpp:489-494
        std::fprintf(stderr, "got profiling automation options without profiling mode, enable it using --profiling <stats>\n");
        return 2;
    }
    if (want_help) scene_path.clear(); // prints the usage
    if (scene_path.empty() || (int)validation + (int)profiling + (int)data_capture != 1 || batch_spp < 1 || width < 1 || height < 1 || (profiling && profiling_frames < 1)) {
        std::fprintf(stderr, "usage: rptr_hip <scene.rpsc> (--validation <prefix> [--validation-spp n] | --profiling <csv prefix> [--profiling-fps f] "
                             "[--profiling-img <prefix>] [--keyframe [len:]file.ini ...] [--profiling-count n] [--synchronous] [--fly-through] [--frames-in-flight n [--frames-per-launch b]] [--animate-wave a k]) [--img w h] [--eye x y z] [--center x y z] "
                             "[--up x y z] [--fov deg] [--variant gltf|diffuse] [--batch-spp k] [--every-frame] [--exr|--pfm|--png] [--config file.ini ...] [--sky-data <dir of the Hosek-Wilkie data headers>]\n"
                             "       rptr_hip <scene.rpsc> --data-capture <prefix> [--data-capture-spp n] [--data-capture-no-rgba] [--data-capture-no-aovs] "
                             "[--data-capture-albedo-roughness] [--data-capture-normal-depth] [--data-capture-motion] [--keyframe ...]   (EXR images per keyframe)\n"
                             "       <scene.vks>: [--remove-first-lods n] [--instance-pruning p] [--small-deformation] [--ignore-animation] [--ignore-textures] "
                             "[--load-specularity] [--dump-scene out.rpsc]\n"
                             "       <scene_file> [<scene_file>...] [--deduplicate-scene] [--merge-partition-instances]\n"
                             "validation, profiling and data-capture mode are mutually exclusive (cmdline.cpp:479-486)\n");
        return 2;
    }
    try {
        rptr::SceneDump scene = load_scene(scene_path);
        // ---- configuration: the scene file's state, then every --config in order, then the command line (main.cpp:121-149, app.cpp:204-213)
        rptr::HostConfig base;
        base.params = scene.render_params;
        base.lighting = scene.lighting;
        base.camera = scene.camera;
        for (const std::string &ini : config_inis) {
            std::printf("Loading config from %s\n", ini.c_str());
            rptr::load_config(ini, base);
        }
        std::vector<rptr::HostConfig> frames; // one per keyframe (profiling mode); empty = the base configuration throughout
        std::vector<double> holds;
        for (const Keyframe &k : keyframes) {
            std::printf("Loading config from %s\n", k.ini.c_str());
            const std::vector<rptr::IniBlock> blocks = rptr::parse_ini(k.ini);
            // every [Application] block of a keyframe file that changes the state is a keyframe; a static file is held for `hold` seconds
            rptr::HostConfig state = frames.empty() ? base : frames.back();
            for (const rptr::IniBlock &b : blocks) rptr::apply_ini_object(b.root, state);
            frames.push_back(state);
            holds.push_back(k.hold);
        }
        // the sky of a scene state (update_config -> update_sky_light, vulkan/render_sky.cpp:25-72): fitted here when a configuration names
        // Sun settings and the Hosek-Wilkie data headers are at hand (--sky-data / RPTR_SKY_DATA; host/sky_fit.hpp)
        rptr::SkyTables sky_tables;
        bool have_sky = false;
        if (sky_data.empty() && std::getenv("RPTR_SKY_DATA")) sky_data = std::getenv("RPTR_SKY_DATA");
        if (!sky_data.empty()) {
            std::string err;
            have_sky = rptr::load_sky_tables(sky_data, sky_tables, err);
            if (!have_sky) throw std::runtime_error("--sky-data: " + err);
            if (!err.empty()) std::fprintf(stderr, "note: %s\n", err.c_str());
        }
        const RptrSceneParams file_sky = scene.scene_params;
        auto sky_of = [&](rptr::HostConfig &c) { // the scene parameters of one state; reference defaults where no file named a value
            RptrSceneParams sp = file_sky;
            if (c.bump_scale > 0.f) sp.normal_z_scale = 1.0f / c.bump_scale; // render_vulkan.cpp:2954-2959
            if (!c.sun_changed) return sp;
            if (!have_sky) {
                const char *note = "Sun settings (height / angle / turbidity / Color) need the Hosek-Wilkie data headers (--sky-data <dir> / RPTR_SKY_DATA): the scene file's sky is kept";
                if (std::find(c.notes.begin(), c.notes.end(), note) == c.notes.end()) c.notes.push_back(note);
                return sp;
            }
            float height, angle;
            rptr::sun_height_angle_from_dir(file_sky.sun_dir, height, angle);
            float dir[3] = {file_sky.sun_dir[0], file_sky.sun_dir[1], file_sky.sun_dir[2]};
            if (!std::isnan(c.sun_height) || !std::isnan(c.sun_angle)) // scene_state.h:85-90
                rptr::sun_dir_from_height_angle(std::isnan(c.sun_height) ? height : c.sun_height, std::isnan(c.sun_angle) ? angle : c.sun_angle, dir);
            const float turbidity = std::isnan(c.turbidity) ? 3.0f : std::min(10.0f, std::max(1.0f, c.turbidity)); // SceneConfig defaults, render_params.glsl.h:157-162
            const float albedo[3] = {std::isnan(c.albedo[0]) ? 0.2f : c.albedo[0], std::isnan(c.albedo[1]) ? 0.2f : c.albedo[1], std::isnan(c.albedo[2]) ? 0.2f : c.albedo[2]};
            rptr::fit_sky(sky_tables, dir, turbidity, albedo, (int)scene.lights.size(), sp);
            return sp;
        };
        scene.scene_params = sky_of(base);
        for (const rptr::HostConfig *c : {&base})
            for (const std::string &n : c->notes) std::fprintf(stderr, "note: %s\n", n.c_str());
        if (base.target_spp > 0 && !validation) target_spp = base.target_spp;
        if (!got_batch_spp) batch_spp = std::max(1, base.params.batch_spp);
        if (!got_variant && base.variant >= 0) variant = base.variant;
        if (upscale >= 1) base.params.render_upscale_factor = upscale;
        if (frames_in_flight == 1) synchronous = true; // (one frame context: nothing to overlap)
        // frame contexts: the reference's two swap buffers; what --frames-in-flight asks for; four for a queued --validation accumulation
        // (its launch sequences are independent of the display: the deeper queue is what fills the GPU)
        const int contexts = std::max(frames_in_flight, (validation && !synchronous && !freeze_frame) ? 4 : (int)rptr::RenderHip::MAX_SWAP_BUFFERS);
        rptr::RenderGroup backend(devices, stripe_rows, contexts, RPTR_CREATE_SET_HW_QUEUES); // (this program owns its process: a hardware queue per frame context)
        backend.initialize(width, height);
        backend.set_scene(scene.desc());
        base.params.batch_spp = batch_spp;
        backend.set_params(base.params, base.lighting);
        backend.update_config(scene.scene_params);
        // render backend options: the point set (with the table its render extension uploads) and the BVH policy of dynamic meshes
        if (rng_variant < 0) rng_variant = base.rng_variant;
        if (rng_variant > 0) {
            const std::string matrices = data_dir() + "/sobol_matrices_1024x32.u32";
            if (rng_variant == RPTR_RNG_VARIANT_BN)
                backend.set_rng_variant(rng_variant, bn_table_path.empty() ? rptr::white_noise_bn_table(matrices) : rptr::read_bn_table(bn_table_path));
            else
                backend.set_rng_variant(rng_variant, rptr::sobol_table(matrices));
        }
        if (force_bvh_rebuild >= 0 || rebuild_triangle_budget >= 0 || base.bvh_policy_set) { // (otherwise the library's default: refit only)
            if (force_bvh_rebuild < 0) force_bvh_rebuild = base.force_bvh_rebuild;
            if (rebuild_triangle_budget < 0) rebuild_triangle_budget = base.rebuild_triangle_budget;
            backend.set_bvh_policy(force_bvh_rebuild != 0, rebuild_triangle_budget);
        }
        scene.camera = base.camera;
        rptr::RenderConfiguration cfg{};
        std::memcpy(cfg.camera.pos, scene.camera.pos, 12);
        std::memcpy(cfg.camera.dir, scene.camera.dir, 12);
        std::memcpy(cfg.camera.up, scene.camera.up, 12);
        cfg.camera.fovy = scene.camera.fovy;
        if (got_eye) std::memcpy(cfg.camera.pos, eye, 12);
        if (got_center || got_eye) { // OrientedCamera: the view direction is center - eye, normalised
            float c[3] = {0, 0, 0};
            if (got_center) std::memcpy(c, center, 12);
            else for (int k = 0; k < 3; ++k) c[k] = scene.camera.pos[k] + scene.camera.dir[k];
            float d[3], len = 0;
            for (int k = 0; k < 3; ++k) { d[k] = c[k] - cfg.camera.pos[k]; len += d[k] * d[k]; }
            len = std::sqrt(len);
            if (len > 0) for (int k = 0; k < 3; ++k) cfg.camera.dir[k] = d[k] / len;
        }
        if (got_up) std::memcpy(cfg.camera.up, up, 12);
        if (fov > 0) cfg.camera.fovy = fov;
        cfg.active_variant = variant;
        cfg.freeze_frame = freeze_frame;
        cfg.reset_accumulation = true; // frame 0 of the accumulation (app.cpp: reset on scene load)
        std::vector<float> img((size_t)width * height * 4);

        if (validation) {
            int accumulated = 0;
            double gpu_ms = 0.0, image_ms = 0.0; // (image_ms: read-back + file)
            const auto v0 = std::chrono::steady_clock::now();
            // The frames of an accumulation are independent launch chains that only meet in the running mean: they are queued as launch
            // sequences of up to 16 samples (rptr_hip_render_batch_async, reset_rest = 0: frame k continues the accumulation of frame k - 1;
            // every frame bit-identical to the frame rendered on its own) with four sequences in flight, and collected in order -- the images
            // are written at the same sample counts, with the same bits, as by the loop of synchronous frames below (--synchronous; a frozen
            // frame repeats its samples and cannot share a sequence).
            if (!synchronous && !freeze_frame && (backend.size() == 1 || !every_frame)) { // (a group's gather assembles the LAST frame of a sequence)
                const int total_frames = (target_spp + batch_spp - 1) / batch_spp;
                // (a launch sequence must fit the sample slots the handle holds: 16 up to ~2.9 Mpixel per rank, fewer above -- 12 at 1440p, 5 at 4K,
                // rptr_hip_get_option "sample_slots"; a frame of more samples than that is split by the library itself, one frame per sequence)
                const int slots = (int)std::max<int64_t>(1, backend.get_option("sample_slots"));
                const int fit = std::max(1, slots / std::max(1, batch_spp));
                const int per_seq = std::min(fit, got_frames_per_launch ? frames_per_launch : std::max(1, std::min(8, 16 / std::max(1, batch_spp))));
                struct Pending {
                    rptr::RenderGroup::Sequence q;
                    int first_frame;
                };
                std::vector<Pending> queue;
                int submitted = 0;
                auto collect_one = [&] {
                    const Pending p = queue.front();
                    queue.erase(queue.begin());
                    for (int k = 0; k < p.q.frames; ++k) {
                        const rptr::RenderStats st = backend.collect_frame(p.q, k);
                        accumulated = st.spp;
                        gpu_ms += st.render_time;
                        if (accumulated >= target_spp || every_frame) {
                            const auto w0 = std::chrono::steady_clock::now();
                            save_image(backend, format, validation_prefix, accumulated, "", width, height, img);
                            image_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                        }
                    }
                };
                while (submitted < total_frames) {
                    const int n = std::min(per_seq, total_frames - submitted);
                    cfg.reset_accumulation = submitted == 0;
                    queue.push_back({backend.submit(cfg, batch_spp, n, /*reset_rest=*/false), submitted});
                    submitted += n;
                    if ((int)queue.size() >= contexts) collect_one();
                }
                while (!queue.empty()) collect_one();
            }
            while (accumulated < target_spp) {
                const rptr::RenderStats st = backend.render(cfg); // params.batch_spp samples
                cfg.reset_accumulation = false;
                accumulated = freeze_frame ? accumulated + batch_spp : st.spp; // (a frozen frame repeats its samples: the application counts, app_state.cpp)
                gpu_ms += st.render_time;
                if (accumulated >= target_spp || every_frame) {
                    const auto w0 = std::chrono::steady_clock::now();
                    save_image(backend, format, validation_prefix, accumulated, "", width, height, img);
                    image_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
                }
            }
            {
                const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - v0).count();
                std::printf("%s: wall %.3f ms rendering + %.3f ms images (read-back and files)\n", backend.name().c_str(), wall - image_ms, image_ms);
            }
            std::printf("%s: %d spp in %.3f ms GPU time -> %s_%04d.%s\n", backend.name().c_str(), accumulated, gpu_ms, validation_prefix.c_str(), accumulated,
                        format == FORMAT_PFM ? "pfm" : format == FORMAT_PNG ? "png" : "exr");
            return 0;
        }
        if (data_capture) {
            // libapp/app_state.cpp:499-531: when a frame is ready (--data-capture-spp samples) the accumulation buffer and the three
            // AOV images
            // of every keyframe (one without --keyframe) are stored as <prefix>_%04d_{rgba,albedo_roughness,normal_depth,motion_jitter}.exr;
            // --data-capture-fps only paces the reference's animation clock between captures (a keyframe here is captured once)
            (void)capture_fps;
            const size_t n_keys = std::max<size_t>(1, frames.size());
            for (size_t k = 0; k < n_keys; ++k) {
                if (!frames.empty()) {
                    rptr::HostConfig st = frames[k];
                    st.params.batch_spp = batch_spp;
                    if (upscale >= 1) st.params.render_upscale_factor = upscale;
                    backend.set_params(st.params, st.lighting);
                    if (st.sun_changed || st.bump_scale > 0.f) backend.update_config(sky_of(st));
                    std::memcpy(cfg.camera.pos, st.camera.pos, 12);
                    std::memcpy(cfg.camera.dir, st.camera.dir, 12);
                    std::memcpy(cfg.camera.up, st.camera.up, 12);
                    if (!got_variant && st.variant >= 0) cfg.active_variant = st.variant;
                    cfg.reset_accumulation = true;
                }
                int accumulated = 0;
                while (accumulated < capture_spp) {
                    const rptr::RenderStats st = backend.render(cfg);
                    cfg.reset_accumulation = false;
                    accumulated = st.spp;
                }
                char idx[16];
                std::snprintf(idx, sizeof(idx), "_%04d", (int)k + 1);
                const std::string pf = capture_prefix + idx;
                if (capture_rgba) save_image(backend, FORMAT_EXR, capture_prefix, (int)k + 1, "_rgba", width, height, img);
                if (capture_albedo) save_aov(backend, rptr::RenderHip::AOVAlbedoRoughnessIndex, pf + "_albedo_roughness", width, height);
                if (capture_normal) save_aov(backend, rptr::RenderHip::AOVNormalDepthIndex, pf + "_normal_depth", width, height);
                if (capture_motion) save_aov(backend, rptr::RenderHip::AOVMotionJitterIndex, pf + "_motion_jitter", width, height);
                std::printf("%s: %d spp -> %s_*.exr\n", backend.name().c_str(), accumulated, pf.c_str());
            }
            return 0;
        }

        // ---- profiling mode
        std::vector<float> rest, cur; // --animate-wave: float positions of geometry 0 at rest
        if (wave_amp != 0.f) {
            if (scene.meshes.empty() || !scene.meshes[0].dynamic) throw std::runtime_error("--animate-wave needs a scene whose mesh 0 is dynamic");
            const RptrGeometryDesc &g = scene.geometries[scene.meshes[0].first_geometry];
            rest.resize((size_t)g.num_tris * 9);
            for (size_t v = 0; v < (size_t)g.num_tris * 3; ++v) { // librender/dequantize.glsl:8-21
                const uint64_t w = g.qpos[v];
                rest[3 * v + 0] = float(uint32_t(w) & 0x1FFFFFu) * g.quantized_scaling[0] + g.quantized_offset[0];
                rest[3 * v + 1] = float(uint32_t(w >> 21) & 0x1FFFFFu) * g.quantized_scaling[1] + g.quantized_offset[1];
                rest[3 * v + 2] = float(uint32_t(w >> 42) & 0x1FFFFFu) * g.quantized_scaling[2] + g.quantized_offset[2];
            }
            cur = rest;
        }
        const std::string csv_path = csv_prefix + ".csv";
        FILE *csv = std::fopen(csv_path.c_str(), "w");
        if (!csv) throw std::runtime_error("cannot open " + csv_path);
        std::fprintf(csv, "frames_total,keyframe,frames_accumulated,render_time_ms,app_time_ms\n");
        const float dt = 1.f / profiling_fps;
        double current_time = 0.0, gpu_ms = 0.0;
        // keyframes: each is held for its length on the animation timeline (default one second = profiling_fps frames); without
        // keyframes the run lasts --profiling-frames frames of the base configuration
        std::vector<double> key_end;
        if (!frames.empty()) {
            double t = 0.0;
            for (double hsec : holds) key_end.push_back(t += std::max(hsec, (double)dt));
            profiling_frames = std::max(1, (int)std::floor(key_end.back() * profiling_fps + 0.5));
        }
        auto last = std::chrono::steady_clock::now();
        // ---- the plan: per frame its keyframe, its view, whether it restarts the accumulation and whether its keyframe ends with it
        // (libapp/app_state.cpp:484-493: an image once per keyframe, at its end; without keyframe files a keyframe is one second)
        struct FramePlan {
            int key = -1;            // index into `frames` (-1: the base configuration)
            int keyframe_no = 1;     // the CSV's keyframe column
            int accumulated = 1;     // the CSV's frames_accumulated column
            bool reset = false, key_ends = false;
            double time = 0.0;
            rptr::RenderCameraParams camera;
        };
        // --fly-through: bench.py's camera path (camera_of): frame k looks along the view yawed by 0.002 rad x s about its up axis from 2 cm x s
        // further to the right, s = (k mod 64) - 32 (64 views symmetric around the configuration's own); evaluated in double as numpy does,
        // rounded to float once
        const rptr::RenderCameraParams base_camera = cfg.camera;
        auto fly_camera = [&](int k) {
            rptr::RenderCameraParams c = base_camera;
            const int s = k % 64 - 32;
            const double a = 0.002 * s;
            const double d[3] = {base_camera.dir[0], base_camera.dir[1], base_camera.dir[2]}, u[3] = {base_camera.up[0], base_camera.up[1], base_camera.up[2]};
            double r[3] = {d[1] * u[2] - d[2] * u[1], d[2] * u[0] - d[0] * u[2], d[0] * u[1] - d[1] * u[0]};
            const double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
            for (double &x : r) x /= rl;
            double nd[3];
            for (int i = 0; i < 3; ++i) nd[i] = std::cos(a) * d[i] + std::sin(a) * r[i];
            const double nl = std::sqrt(nd[0] * nd[0] + nd[1] * nd[1] + nd[2] * nd[2]);
            for (int i = 0; i < 3; ++i) {
                c.pos[i] = (float)(double(base_camera.pos[i]) + 0.02 * s * r[i]);
                c.dir[i] = (float)(nd[i] / nl);
            }
            return c;
        };
        std::vector<FramePlan> plan((size_t)profiling_frames);
        {
            int active_key = -1, acc = 0;
            rptr::RenderCameraParams cam = cfg.camera;
            for (int frame = 0; frame < profiling_frames; ++frame) {
                FramePlan &p = plan[(size_t)frame];
                p.reset = frame == 0;
                if (!frames.empty()) { // the keyframe this frame belongs to; a change applies its state and restarts the accumulation
                    int k = 0;
                    while (k + 1 < (int)key_end.size() && current_time >= key_end[(size_t)k] - 1e-9) ++k;
                    if (k != active_key) {
                        active_key = k;
                        const rptr::HostConfig &st = frames[(size_t)k];
                        std::memcpy(cam.pos, st.camera.pos, 12);
                        std::memcpy(cam.dir, st.camera.dir, 12);
                        std::memcpy(cam.up, st.camera.up, 12);
                        p.reset = true;
                    }
                }
                if (!rest.empty()) p.reset = true; // the geometry moves: the accumulation starts over
                p.camera = cam;
                if (fly_through) { // a moved camera restarts the accumulation (the application resets when the view changes)
                    p.camera = fly_camera(frame);
                    p.reset = true;
                }
                p.key = active_key;
                p.time = current_time;
                acc = p.reset ? 1 : acc + 1;
                p.accumulated = acc;
                p.keyframe_no = frames.empty() ? (int)std::floor(current_time) + 1 : active_key + 1;
                p.key_ends = frames.empty() ? (current_time + dt) >= std::ceil(current_time + 1e-9) : (current_time + dt) >= key_end[(size_t)active_key] - 1e-9;
                if (!freeze_frame) current_time += dt; // --freeze-frame keeps repeating the same frame (app.cpp:339-345)
                else if (!frames.empty() && frame + 1 >= (int)std::floor(key_end[(size_t)active_key] * profiling_fps + 0.5)) current_time = key_end[(size_t)active_key];
            }
        }
        int applied_key = -1;
        auto apply_key = [&](int k) { // the state of keyframe k: parameters, sky, variant (what is submitted from here on uses it)
            if (k < 0 || k == applied_key) return;
            applied_key = k;
            rptr::HostConfig st = frames[(size_t)k];
            st.params.batch_spp = batch_spp;
            if (upscale >= 1) st.params.render_upscale_factor = upscale;
            backend.set_params(st.params, st.lighting);
            if (st.sun_changed || st.bump_scale > 0.f) backend.update_config(sky_of(st));
            if (!got_variant && st.variant >= 0) cfg.active_variant = st.variant;
        };
        auto config_of = [&](const FramePlan &p) {
            rptr::RenderConfiguration c = cfg;
            c.camera = p.camera;
            c.reset_accumulation = p.reset;
            c.time = p.time;
            return c;
        };
        // ---- queued deeper than the reference's two swap buffers (--frames-in-flight n [--frames-per-launch b]): launch sequences of up to b
        // frames of one keyframe, every frame with its own view (rptr_hip_render_batch_cameras_async), n sequences in flight -- the schedule
        // bench.py times (the library asks the HIP runtime for a hardware queue per frame context itself: csrc/rptr_hip.hip
        // ensure_hw_queues). One CSV row per frame as it is collected; render_time_ms is the frame's share of its sequence.
        if (frames_in_flight > 1 && rest.empty() && !freeze_frame) {
            frames_per_launch = std::min(frames_per_launch, std::max(1, 16 / std::max(1, batch_spp)));
            struct Pending {
                rptr::RenderGroup::Sequence q;
                int first;
            };
            std::vector<Pending> queue;
            apply_key(plan[0].key);
            for (int w = 0; w < 2; ++w) { // warm-up: every context renders once, the adaptive tail hand-over settles
                std::vector<rptr::RenderGroup::Sequence> warm;
                for (int k = 0; k < frames_in_flight; ++k) {
                    rptr::RenderConfiguration c = config_of(plan[0]);
                    c.reset_accumulation = true;
                    warm.push_back(backend.submit(c, batch_spp, frames_per_launch));
                }
                for (const auto &q : warm) backend.collect(q);
            }
            const auto t0 = std::chrono::steady_clock::now();
            const double rays0 = backend.rays_traced();
            last = t0;
            int submitted = 0, collected = 0;
            auto drain_one = [&] {
                const Pending p = queue.front();
                queue.erase(queue.begin());
                for (int k = 0; k < p.q.frames; ++k) {
                    const FramePlan &fp = plan[(size_t)(p.first + k)];
                    const rptr::RenderStats st = backend.collect_frame(p.q, k);
                    gpu_ms += st.render_time;
                    const auto now = std::chrono::steady_clock::now();
                    std::fprintf(csv, "%d,%d,%d,%g,%g\n", ++collected, fp.keyframe_no, fp.accumulated, st.render_time, std::chrono::duration<double, std::milli>(now - last).count());
                    last = now;
                    // (an image per keyframe: of a one-device group, or of the frame a group's gather assembled last)
                    if (!profiling_img_prefix.empty() && fp.key_ends && (backend.size() == 1 || k == p.q.frames - 1))
                        save_image(backend, format, profiling_img_prefix, fp.keyframe_no, "", width, height, img);
                }
            };
            while (submitted < profiling_frames) {
                // the next sequence: frames of ONE keyframe whose later frames all restart the accumulation or all continue it
                int n = 1;
                const FramePlan &f0 = plan[(size_t)submitted];
                while (n < frames_per_launch && submitted + n < profiling_frames && plan[(size_t)(submitted + n)].key == f0.key &&
                       plan[(size_t)(submitted + n)].reset == plan[(size_t)(submitted + 1)].reset && !(n >= 1 && plan[(size_t)(submitted + n - 1)].key_ends && !profiling_img_prefix.empty() && backend.size() > 1))
                    ++n;
                apply_key(f0.key);
                std::vector<rptr::RenderConfiguration> cs;
                for (int k = 0; k < n; ++k) cs.push_back(config_of(plan[(size_t)(submitted + k)]));
                queue.push_back({backend.submit(cs, batch_spp, n > 1 ? plan[(size_t)(submitted + 1)].reset : true), submitted});
                submitted += n;
                if ((int)queue.size() >= frames_in_flight) drain_one();
            }
            while (!queue.empty()) drain_one();
            const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            std::fclose(csv);
            std::printf("%s: %d frames, %d in flight x %d per launch sequence%s: %.4f ms per frame (wall), %.1f Mrays/s -> %s\n", backend.name().c_str(), profiling_frames,
                        frames_in_flight, frames_per_launch, fly_through ? ", a camera per frame" : "", wall_ms / profiling_frames, (backend.rays_traced() - rays0) / wall_ms * 1e-3,
                        csv_path.c_str());
            return 0;
        }
        // ---- the reference's loop (app.cpp:453-469): begin_frame / draw_frame / end_frame per frame, with the display's command stream (two
        // frames in flight, statistics two frames late) or, --synchronous, without one
        rptr::CommandStream display_stream;
        rptr::CommandStream *const cmd_stream = synchronous ? nullptr : &display_stream;
        if (!rest.empty() || fly_through) { // (a short warm-up when the run is a measurement: the adaptive tail hand-over settles in two frames)
            apply_key(plan[0].key);
            for (int w = 0; w < 4; ++w) {
                rptr::RenderConfiguration c = config_of(plan[0]);
                c.reset_accumulation = true;
                backend.frame(cmd_stream, c);
            }
            backend.flush_pipeline();
        }
        const auto loop_t0 = std::chrono::steady_clock::now();
        const double loop_rays0 = backend.rays_traced();
        last = loop_t0;
        for (int frame = 0; frame < profiling_frames; ++frame) {
            const FramePlan &fp = plan[(size_t)frame];
            apply_key(fp.key);
            if (!rest.empty()) { // the geometry moves: new vertices, refit, and the accumulation starts over
                const float phase = 6.283185307179586f * (float)fp.time;
                for (size_t v = 0; v < rest.size() / 3; ++v) cur[3 * v + 1] = rest[3 * v + 1] + wave_amp * std::sin(wave_k * rest[3 * v] + phase);
                backend.update_vertices(scene.meshes[0].first_geometry, cur.data(), (uint32_t)(cur.size() / 3));
                backend.refit();
            }
            const rptr::RenderStats st = backend.frame(cmd_stream, config_of(fp));
            gpu_ms += st.render_time;
            const auto now = std::chrono::steady_clock::now();
            const double app_ms = std::chrono::duration<double, std::milli>(now - last).count();
            last = now;
            std::fprintf(csv, "%d,%d,%d,%g,%g\n", frame + 1, fp.keyframe_no, fp.accumulated, st.render_time, app_ms);
            if (!profiling_img_prefix.empty() && fp.key_ends) save_image(backend, format, profiling_img_prefix, fp.keyframe_no, "", width, height, img); // (a read-back finishes what is in flight)
        }
        backend.flush_pipeline();
        {
            const double wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - loop_t0).count();
            std::printf("%s: frame loop %s: %.4f ms per frame (wall), %.1f Mrays/s\n", backend.name().c_str(),
                        cmd_stream ? "through a command stream (two frames in flight)" : "synchronous", wall_ms / profiling_frames,
                        (backend.rays_traced() - loop_rays0) / wall_ms * 1e-3);
        }
        std::fclose(csv);
        std::printf("%s: %d frames at %.3g fps animation time, %.3f ms GPU time per frame -> %s\n", backend.name().c_str(), profiling_frames, profiling_fps,
                    gpu_ms / profiling_frames, csv_path.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rptr_hip: %s\n", e.what());
        return 3;
    }
    return 0;
}
