// demo_host.cpp -- smallest C++ client of the RenderBackend-shaped host class: one triangle
// above a ground quad, 64x64, 2 spp, prints the mean radiance. Exits 3 with the backend's
// message when no GPU is present (the product has no CPU fallback).
#include "render_hip.hpp"

#include <cstdio>
#include <cstring>
#include <dlfcn.h>

static uint64_t qpos(float x, float y, float z, const float lo[3], const float ext[3]) { // librender/quantize.h:7-11
    const float p[3] = {(x - lo[0]) * 2097152.0f / ext[0], (y - lo[1]) * 2097152.0f / ext[1], (z - lo[2]) * 2097152.0f / ext[2]};
    uint64_t u[3];
    for (int k = 0; k < 3; ++k) u[k] = (uint64_t)(p[k] < 0 ? 0 : (p[k] > 2097151.0f ? 2097151.0f : p[k]));
    return u[0] | (u[1] << 21) | (u[2] << 42);
}

int main() {
    try {
        rptr::RenderHip backend;
        std::printf("backend: %s\n", backend.name().c_str());
        const float lo[3] = {-2, -1, -2}, ext[3] = {4, 2, 4};
        const float v[9][3] = {{-2, -1, -2}, {-2, -1, 2}, {2, -1, 2}, {-2, -1, -2}, {2, -1, 2}, {2, -1, -2}, {-1, 0, 0}, {1, 0, 0}, {0, 1, 0}};
        uint64_t q[9];
        for (int i = 0; i < 9; ++i) q[i] = qpos(v[i][0], v[i][1], v[i][2], lo, ext);
        RptrGeometryDesc g{};
        g.qpos = q;
        g.num_tris = 3;
        for (int k = 0; k < 3; ++k) {
            g.quantized_scaling[k] = ext[k] / 2097152.0f;                  // quantize.h:13-15
            g.quantized_offset[k] = lo[k] + ext[k] * 0.5f / 2097152.0f;    // quantize.h:16-18
        }
        RptrMeshDesc mesh{0, 1, 0};
        const int32_t mat_off[1] = {0};
        RptrParameterizedMeshDesc pm{0, mat_off, nullptr};
        RptrInstanceDesc inst{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}, 0};
        RptrBaseMaterial m{};
        m.base_color[0] = m.base_color[1] = m.base_color[2] = 0.8f;
        m.normal_map = -1;
        m.flags = RPTR_BASE_MATERIAL_NOALPHA;
        m.roughness = 1.0f;
        m.specular = 0.5f;
        m.ior = 1.5f;
        RptrSceneDesc scene{&g, 1, &mesh, 1, &pm, 1, &inst, 1, &m, 1, nullptr, 0};
        backend.initialize(64, 64);
        backend.set_scene(scene);
        RptrSceneParams sp{}; // a bare sun (no sky fit here: that is the reference's Hosek code)
        sp.sun_dir[0] = 0.30151134f; sp.sun_dir[1] = 0.80403025f; sp.sun_dir[2] = 0.50251891f;
        sp.sun_cos_angle = 0.99998933f;
        sp.sun_radiance[0] = 25000.f; sp.sun_radiance[1] = 21000.f; sp.sun_radiance[2] = 16000.f; sp.sun_radiance[3] = 1.0f;
        sp.normal_z_scale = 1.0f;
        backend.update_config(sp);
        rptr::RenderConfiguration cfg{};
        const float pos[3] = {0, 0.5f, 4}, dir[3] = {0, -0.1f, -1}, up[3] = {0, 1, 0};
        std::memcpy(cfg.camera.pos, pos, 12);
        std::memcpy(cfg.camera.dir, dir, 12);
        std::memcpy(cfg.camera.up, up, 12);
        cfg.camera.fovy = 45;
        cfg.reset_accumulation = true;
        cfg.active_variant = RPTR_VARIANT_SIMPLE;
        rptr::RenderStats st = backend.render(cfg, 2);
        std::vector<float> img(64 * 64 * 4);
        if (backend.readback_framebuffer(img.size(), img.data()) != img.size()) return 2;
        double sum = 0, alpha = 0;
        for (size_t i = 0; i < img.size(); i += 4) {
            sum += img[i] + img[i + 1] + img[i + 2];
            alpha += img[i + 3];
        }
        std::printf("spp %d  %.3f ms  mean radiance %.4f  coverage %.3f\n", st.spp, st.render_time, sum / (3 * 64 * 64), alpha / (64 * 64));
        if (!(sum > 0 && alpha > 0)) return 4;
        // ---- ray queries as RenderBackend::enable_ray_queries / render_ray_queries run them: queries in a DEVICE buffer the backend owns
        // (some other device pass writes them in the reference; this demo copies three in), results in its result buffer. The HIP runtime
        // is in the process already (the backend links it): hipMemcpy by name, so that this file needs no HIP headers.
        typedef int (*memcpy_fn)(void *, const void *, size_t, int);
        memcpy_fn hip_memcpy = (memcpy_fn)dlsym(RTLD_DEFAULT, "hipMemcpy");
        RptrRenderRayQuery rq[3] = {};
        const float ro[3][3] = {{0, 0.3f, 4}, {0, 5, 0.5f}, {0, 0.5f, 4}}, rd[3][3] = {{0, 0, -1}, {0, -1, 0}, {0, 1, 0}}; // the triangle, the ground, the sky
        for (int i = 0; i < 3; ++i) {
            std::memcpy(rq[i].origin, ro[i], 12);
            std::memcpy(rq[i].dir, rd[i], 12);
            rq[i].t_max = 1e20f;
        }
        float host_res[12], dev_res[12], rt_res[12];
        if (!backend.render_ray_queries(rq, 3, host_res)) return 5;
        backend.enable_ray_queries(16);
        if (!hip_memcpy || hip_memcpy(backend.ray_query_buffer(), rq, sizeof rq, 1 /* hipMemcpyHostToDevice */) != 0) return 6;
        if (!backend.render_ray_queries(3)) return 7;
        if (hip_memcpy(dev_res, backend.ray_result_buffer(), sizeof dev_res, 2 /* hipMemcpyDeviceToHost: after the backend's stream work */) != 0) return 8;
        // (hipMemcpy on the NULL stream does not wait for a non-blocking stream: the backend's next synchronous call does)
        float again[12];
        backend.render_ray_queries(rq, 3, again);
        if (hip_memcpy(dev_res, backend.ray_result_buffer(), sizeof dev_res, 2) != 0) return 8;
        const bool same = std::memcmp(host_res, dev_res, sizeof host_res) == 0;
        int32_t prim0, prim1, prim2;
        std::memcpy(&prim0, &host_res[3], 4);
        std::memcpy(&prim1, &host_res[7], 4);
        std::memcpy(&prim2, &host_res[11], 4);
        std::printf("ray queries: device buffers %s host arrays; primitives %d %d %d\n", same ? "=" : "!=", prim0, prim1, prim2);
        if (!same || prim0 != 2 || prim1 < 0 || prim1 > 1 || prim2 != -1) return 9;
        // ---- the ray-query-only surface (struct RaytraceBackend, librender/raytrace_backend.h:13-19)
        {
            rptr::RaytraceHip rt;
            rt.set_scene(scene);
            const bool ok = rt.trace_ray(rq, 3, rt_res) == 3 && std::memcmp(rt_res, host_res, sizeof host_res) == 0;
            std::printf("%s: %s\n", rt.name().c_str(), ok ? "same hits" : "DIFFERENT hits");
            if (!ok) return 10;
        }
        // ---- light_sampling_variant NONE (rendering/mc/nee.glsl:12-14): every NEE sample goes to the sun; this scene has no emitters, so
        // the image must not change
        std::vector<float> img2(img.size());
        {
            rptr::RenderHip b2; // (a fresh handle: the seeds of a frame depend on how many frames its handle has reset before)
            b2.initialize(64, 64);
            b2.set_scene(scene);
            b2.update_config(sp);
            b2.set_light_sampling_variant(0);
            b2.render(cfg, 2);
            b2.readback_framebuffer(img2.size(), img2.data());
        }
        std::printf("light sampling NONE: image %s\n", std::memcmp(img.data(), img2.data(), img.size() * 4) == 0 ? "unchanged" : "CHANGED");
        if (std::memcmp(img.data(), img2.data(), img.size() * 4) != 0) return 11;
        // ---- the reference's frame loop (app.cpp:453-469): begin_frame / draw_frame / end_frame with the application's CommandStream*.
        // Eight frames, the camera moves for the first five (every move restarts the accumulation), then three frames accumulate: through a
        // command stream (asynchronous: two frames in flight, statistics two frames late) and with nullptr (synchronous) -- same images.
        auto loop = [&](bool asynchronous, std::vector<float> &out5, std::vector<float> &out8, int &delay_seen, int &valid_after, int &spp_end) {
            rptr::RenderHip b;
            b.initialize(64, 64);
            b.set_scene(scene);
            b.update_config(sp);
            b.params.batch_spp = 2;
            rptr::CommandStream display_stream;
            rptr::CommandStream *cs = asynchronous ? &display_stream : nullptr;
            delay_seen = 0;
            valid_after = -1;
            for (int f = 0; f < 8; ++f) {
                rptr::RenderConfiguration c = cfg;
                c.camera.pos[0] = 0.05f * float(f < 5 ? f : 4);
                c.reset_accumulation = f <= 4;
                c.active_swap_buffer_count = asynchronous ? -1 : 1; // app.cpp:386-389
                b.begin_frame(cs, c);
                b.draw_frame(cs, c.active_variant);
                b.end_frame(cs, c.active_variant);
                const rptr::RenderStats s = b.stats();
                if (s.has_valid_frame_stats && valid_after < 0) valid_after = f;
                delay_seen = s.frame_stats_delay;
                spp_end = s.spp;
                if (f == 4) {
                    out5.resize(img.size());
                    if (b.readback_framebuffer(out5.size(), out5.data()) != out5.size()) return false;
                }
            }
            out8.resize(img.size());
            return b.readback_framebuffer(out8.size(), out8.data()) == out8.size();
        };
        std::vector<float> a5, a8, s5, s8;
        int a_delay = 0, a_valid = 0, a_spp = 0, s_delay = 0, s_valid = 0, s_spp = 0;
        if (!loop(true, a5, a8, a_delay, a_valid, a_spp) || !loop(false, s5, s8, s_delay, s_valid, s_spp)) return 12;
        const bool loop_same = std::memcmp(a5.data(), s5.data(), a5.size() * 4) == 0 && std::memcmp(a8.data(), s8.data(), a8.size() * 4) == 0;
        std::printf("frame loop through a CommandStream: images %s the synchronous loop's; frame_stats_delay %d / %d, first valid stats at frame %d / %d, spp %d / %d\n",
                    loop_same ? "=" : "!=", a_delay, s_delay, a_valid, s_valid, a_spp, s_spp);
        return loop_same && a_delay == 2 && s_delay == 0 && a_valid == 2 && s_valid == 0 && a_spp == 8 && s_spp == 8 ? 0 : 13;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
