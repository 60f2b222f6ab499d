// demo_host.cpp -- smallest C++ client of the RenderBackend-shaped host class: one triangle
// above a ground quad, 64x64, 2 spp, prints the mean radiance. Exits 3 with the backend's
// message when no GPU is present (the product has no CPU fallback).
#include "render_hip.hpp"

#include <cstdio>
#include <cstring>

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
        return (sum > 0 && alpha > 0) ? 0 : 4;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 3;
    }
}
