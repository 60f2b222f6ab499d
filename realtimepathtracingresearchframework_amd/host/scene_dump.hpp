// scene_dump.hpp -- reads the flat scene file that scenes.py: Scene.dump() writes and exposes it as RptrSceneDesc.
// It stands in for the reference's scene loaders (.vks, ext/libvkr) in the C++ host tools: the adapter of
// INTEGRATION.md fills the same RptrSceneDesc from librender's `Scene` instead.
//
// Layout (little-endian): "RPSC1\0\0\0"; u32 counts: geometries, meshes, parameterized meshes, instances, materials,
// lights; per geometry: u32 num_tris, has_normals, has_uvs; f32 quantized_scaling[3], quantized_offset[3]; u32
// has_attribute_stream; u64 qpos[3*num_tris]; u64 qnrm_uv[3*num_tris] if present; per mesh: u32 first_geometry,
// num_geometries, dynamic; per parameterized mesh: u32 mesh, n; i32 material_offsets[n]; u32 n_ids; u8 ids[n_ids];
// per instance: f32 transform[12], u32 parameterized_mesh; RptrBaseMaterial[]; RptrTriLightData[]; then RptrCamera,
// RptrSceneParams, RptrRenderParams, RptrLightSamplingConfig; optionally u32 num_textures and per texture u32 width,
// height, srgb + width*height RGBA8 texels.
#pragma once
#include "../../include/rptr_hip.h"

#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace rptr {

struct SceneDump {
    std::vector<std::vector<uint64_t>> qpos, qnu;
    std::vector<RptrGeometryDesc> geometries;
    std::vector<RptrMeshDesc> meshes;
    std::vector<std::vector<int32_t>> offsets;
    std::vector<std::vector<uint8_t>> tri_ids;
    std::vector<RptrParameterizedMeshDesc> pmeshes;
    std::vector<RptrInstanceDesc> instances;
    std::vector<RptrBaseMaterial> materials;
    std::vector<RptrTriLightData> lights;
    RptrCamera camera{};
    RptrSceneParams scene_params{};
    RptrRenderParams render_params{};
    RptrLightSamplingConfig lighting{};
    std::vector<std::vector<uint8_t>> texels;
    std::vector<RptrTextureDesc> textures;

    RptrSceneDesc desc() const {
        RptrSceneDesc d{};
        d.geometries = geometries.data();
        d.num_geometries = (uint32_t)geometries.size();
        d.meshes = meshes.data();
        d.num_meshes = (uint32_t)meshes.size();
        d.parameterized_meshes = pmeshes.data();
        d.num_parameterized_meshes = (uint32_t)pmeshes.size();
        d.instances = instances.data();
        d.num_instances = (uint32_t)instances.size();
        d.materials = materials.data();
        d.num_materials = (uint32_t)materials.size();
        d.lights = lights.empty() ? nullptr : lights.data();
        d.num_lights = (uint32_t)lights.size();
        d.textures = textures.empty() ? nullptr : textures.data();
        d.num_textures = (uint32_t)textures.size();
        return d;
    }

    // the same layout back to a file (rptr_cli --dump-scene: what a scene read from a .vks file looks like to the backend)
    void save(const std::string &path) const {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot write " + path);
        auto wr = [&](const void *src, size_t n) {
            if (n && std::fwrite(src, 1, n, f) != n) {
                std::fclose(f);
                throw std::runtime_error("short write " + path);
            }
        };
        wr("RPSC1\0\0\0", 8);
        const uint32_t n[6] = {(uint32_t)geometries.size(), (uint32_t)meshes.size(), (uint32_t)pmeshes.size(), (uint32_t)instances.size(),
                               (uint32_t)materials.size(), (uint32_t)lights.size()};
        wr(n, sizeof(n));
        for (const RptrGeometryDesc &g : geometries) {
            const uint32_t h[3] = {g.num_tris, g.has_normals, g.has_uvs}, has_attr = g.qnrm_uv ? 1u : 0u;
            wr(h, sizeof(h));
            wr(g.quantized_scaling, 12);
            wr(g.quantized_offset, 12);
            wr(&has_attr, 4);
            wr(g.qpos, (size_t)3 * g.num_tris * 8);
            if (g.qnrm_uv) wr(g.qnrm_uv, (size_t)3 * g.num_tris * 8);
        }
        for (const RptrMeshDesc &m : meshes) {
            const uint32_t h[3] = {m.first_geometry, m.num_geometries, m.dynamic};
            wr(h, sizeof(h));
        }
        for (size_t i = 0; i < pmeshes.size(); ++i) {
            const uint32_t h[2] = {pmeshes[i].mesh, (uint32_t)offsets[i].size()}, nid = pmeshes[i].tri_material_ids ? (uint32_t)tri_ids[i].size() : 0u;
            wr(h, sizeof(h));
            wr(offsets[i].data(), offsets[i].size() * 4);
            wr(&nid, 4);
            wr(tri_ids[i].data(), nid);
        }
        for (const RptrInstanceDesc &in : instances) {
            wr(in.transform, 48);
            wr(&in.parameterized_mesh, 4);
        }
        wr(materials.data(), materials.size() * sizeof(RptrBaseMaterial));
        wr(lights.data(), lights.size() * sizeof(RptrTriLightData));
        wr(&camera, sizeof(camera));
        wr(&scene_params, sizeof(scene_params));
        wr(&render_params, sizeof(render_params));
        wr(&lighting, sizeof(lighting));
        const uint32_t ntex = (uint32_t)textures.size();
        wr(&ntex, 4);
        for (size_t i = 0; i < textures.size(); ++i) {
            const uint32_t h3[3] = {textures[i].width, textures[i].height, (textures[i].srgb ? 1u : 0u) | ((textures[i].mip_levels > 1u ? textures[i].mip_levels : 0u) << 8)};
            wr(h3, sizeof(h3));
            wr(texels[i].data(), texels[i].size());
        }
        std::fclose(f);
    }

    static SceneDump load(const std::string &path) {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        auto rd = [&](void *dst, size_t n) {
            if (n && std::fread(dst, 1, n, f) != n) {
                std::fclose(f);
                throw std::runtime_error("truncated scene file " + path);
            }
        };
        char magic[8];
        rd(magic, 8);
        if (std::memcmp(magic, "RPSC1\0\0\0", 8) != 0) {
            std::fclose(f);
            throw std::runtime_error(path + " is not a scene dump");
        }
        uint32_t n[6];
        rd(n, sizeof(n));
        SceneDump s;
        s.qpos.resize(n[0]);
        s.qnu.resize(n[0]);
        s.geometries.resize(n[0]);
        for (uint32_t i = 0; i < n[0]; ++i) {
            uint32_t h[3], has_attr;
            RptrGeometryDesc &g = s.geometries[i];
            std::memset(&g, 0, sizeof(g));
            rd(h, sizeof(h));
            rd(g.quantized_scaling, 12);
            rd(g.quantized_offset, 12);
            rd(&has_attr, 4);
            g.num_tris = h[0];
            g.has_normals = h[1];
            g.has_uvs = h[2];
            s.qpos[i].resize((size_t)3 * h[0]);
            rd(s.qpos[i].data(), s.qpos[i].size() * 8);
            g.qpos = s.qpos[i].data();
            if (has_attr) {
                s.qnu[i].resize((size_t)3 * h[0]);
                rd(s.qnu[i].data(), s.qnu[i].size() * 8);
                g.qnrm_uv = s.qnu[i].data();
            }
        }
        s.meshes.resize(n[1]);
        for (RptrMeshDesc &m : s.meshes) {
            uint32_t h[3];
            rd(h, sizeof(h));
            m.first_geometry = h[0];
            m.num_geometries = h[1];
            m.dynamic = h[2];
        }
        s.offsets.resize(n[2]);
        s.tri_ids.resize(n[2]);
        s.pmeshes.resize(n[2]);
        for (uint32_t i = 0; i < n[2]; ++i) {
            uint32_t h[2], nid;
            rd(h, sizeof(h));
            s.offsets[i].resize(h[1]);
            rd(s.offsets[i].data(), (size_t)h[1] * 4);
            rd(&nid, 4);
            s.tri_ids[i].resize(nid);
            rd(s.tri_ids[i].data(), nid);
            s.pmeshes[i].mesh = h[0];
            s.pmeshes[i].material_offsets = s.offsets[i].data();
            s.pmeshes[i].tri_material_ids = nid ? s.tri_ids[i].data() : nullptr;
        }
        s.instances.resize(n[3]);
        for (RptrInstanceDesc &in : s.instances) {
            rd(in.transform, 48);
            rd(&in.parameterized_mesh, 4);
        }
        s.materials.resize(n[4]);
        rd(s.materials.data(), (size_t)n[4] * sizeof(RptrBaseMaterial));
        s.lights.resize(n[5]);
        rd(s.lights.data(), (size_t)n[5] * sizeof(RptrTriLightData));
        rd(&s.camera, sizeof(s.camera));
        rd(&s.scene_params, sizeof(s.scene_params));
        rd(&s.render_params, sizeof(s.render_params));
        rd(&s.lighting, sizeof(s.lighting));
        uint32_t ntex = 0; // optional trailing section: u32 count, per texture u32 width, height, srgb + RGBA8 rows
        if (std::fread(&ntex, 4, 1, f) == 1 && ntex) {
            s.texels.resize(ntex);
            s.textures.resize(ntex);
            for (uint32_t i = 0; i < ntex; ++i) {
                uint32_t h3[3];
                rd(h3, sizeof(h3));
                const uint32_t levels = h3[2] >> 8; // srgb flag in the low byte, number of mip levels above it (0 = level 0 only)
                size_t texels = 0;
                for (uint32_t l = 0, w = h3[0], hh = h3[1]; l < std::max(1u, levels); ++l, w = std::max(1u, w / 2), hh = std::max(1u, hh / 2)) texels += (size_t)w * hh;
                s.texels[i].resize(texels * 4);
                rd(s.texels[i].data(), s.texels[i].size());
                s.textures[i].rgba8 = s.texels[i].data();
                s.textures[i].width = h3[0];
                s.textures[i].height = h3[1];
                s.textures[i].srgb = h3[2] & 0xFFu;
                s.textures[i].mip_levels = levels;
            }
        }
        std::fclose(f);
        return s;
    }
};

} // namespace rptr
