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

#include <algorithm>
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

    // the descriptors point into the owned storage (call after anything that moved or re-indexed it)
    void fix_pointers() {
        for (size_t i = 0; i < geometries.size(); ++i) {
            geometries[i].qpos = qpos[i].data();
            geometries[i].qnrm_uv = qnu[i].empty() ? nullptr : qnu[i].data();
        }
        for (size_t i = 0; i < pmeshes.size(); ++i) {
            pmeshes[i].material_offsets = offsets[i].data();
            pmeshes[i].tri_material_ids = tri_ids[i].empty() ? nullptr : tri_ids[i].data();
        }
        for (size_t i = 0; i < textures.size(); ++i) textures[i].rgba8 = texels[i].data();
    }

    // A parameter of a material is a literal or a texture handle (rendering/bsdfs/texture_channel_mask.h: sign bit, channel in bits
    // 29..30, texture index in bits 0..28); normal_map is a plain index (< 0: none).
    template <class F>
    static void for_each_texture_ref(RptrBaseMaterial &m, F &&f) {
        float *fields[] = {&m.base_color[0], &m.roughness, &m.specular, &m.metallic, &m.sheen, &m.sheen_tint, &m.clearcoat, &m.clearcoat_gloss, &m.ior,
                           &m.specular_transmission, &m.anisotropy, &m.specular_tint, &m.transmission_color[0], &m.emission_intensity};
        for (float *p : fields) {
            uint32_t bits;
            std::memcpy(&bits, p, 4);
            if (bits & 0x80000000u) {
                uint32_t id = bits & 0x1FFFFFFFu;
                f(id);
                bits = (bits & 0xE0000000u) | (id & 0x1FFFFFFFu);
                std::memcpy(p, &bits, 4);
            }
        }
        if (m.normal_map >= 0) {
            uint32_t id = (uint32_t)m.normal_map;
            f(id);
            m.normal_map = (int32_t)id;
        }
    }

    // Scene::Scene(fnames, ...) (librender/scene.cpp:50-69): every further scene file appends its meshes, parameterized meshes,
    // instances, materials and textures behind what is there, indices shifted. Camera and parameters stay those of the first file;
    // the lights are collected again by the caller (they are binned over the whole scene).
    void append(const SceneDump &o) {
        const uint32_t g0 = (uint32_t)geometries.size(), m0 = (uint32_t)meshes.size(), p0 = (uint32_t)pmeshes.size(), mat0 = (uint32_t)materials.size(),
                       t0 = (uint32_t)textures.size();
        qpos.insert(qpos.end(), o.qpos.begin(), o.qpos.end());
        qnu.insert(qnu.end(), o.qnu.begin(), o.qnu.end());
        geometries.insert(geometries.end(), o.geometries.begin(), o.geometries.end());
        for (RptrMeshDesc m : o.meshes) {
            m.first_geometry += g0;
            meshes.push_back(m);
        }
        for (size_t i = 0; i < o.pmeshes.size(); ++i) {
            RptrParameterizedMeshDesc pm = o.pmeshes[i];
            pm.mesh += m0;
            std::vector<int32_t> off = o.offsets[i];
            for (int32_t &v : off) v += (int32_t)mat0;
            offsets.push_back(std::move(off));
            tri_ids.push_back(o.tri_ids[i]);
            pmeshes.push_back(pm);
        }
        for (RptrInstanceDesc in : o.instances) {
            in.parameterized_mesh += p0;
            instances.push_back(in);
        }
        for (RptrBaseMaterial m : o.materials) {
            for_each_texture_ref(m, [&](uint32_t &id) { id += t0; });
            materials.push_back(m);
        }
        texels.insert(texels.end(), o.texels.begin(), o.texels.end());
        textures.insert(textures.end(), o.textures.begin(), o.textures.end());
        fix_pointers();
    }

    // SceneLoaderParams::PerFile::merge_partition_instances (librender/scene.cpp:757-797): a scene exported in partitions has runs of
    // consecutive instances with the SAME transform, each with a mesh of its own; such a run becomes ONE instance whose mesh lists the
    // geometries of all of them (material offsets appended alike). Instances whose parameterized mesh carries per-triangle material ids
    // are left alone and end a run (the reference's condition reads `lod_group == 0 || ... && !per_triangle_materials()`, which by
    // operator precedence would also merge those -- without their ids; the intended rule is implemented). Instance indices after the first
    // merged run shift, as in the reference. Returns the number of instances merged away.
    size_t merge_partition_instances(size_t instance_base = 0) {
        std::vector<std::vector<uint32_t>> mesh_geoms(meshes.size());
        for (size_t m = 0; m < meshes.size(); ++m)
            for (uint32_t j = 0; j < meshes[m].num_geometries; ++j) mesh_geoms[m].push_back(meshes[m].first_geometry + j);
        bool have_cursor = false;
        size_t cursor_i = instance_base, ic = instance_base;
        float cursor_transform[12];
        for (size_t i = instance_base; i < instances.size(); ++i) {
            const uint32_t p = instances[i].parameterized_mesh;
            if (!tri_ids[p].empty()) {
                have_cursor = false;
                instances[ic++] = instances[i];
                continue;
            }
            const uint32_t cp = instances[cursor_i].parameterized_mesh;
            const bool merge = have_cursor && std::memcmp(cursor_transform, instances[i].transform, 48) == 0 && meshes[pmeshes[p].mesh].dynamic == meshes[pmeshes[cp].mesh].dynamic &&
                               pmeshes[p].mesh != pmeshes[cp].mesh;
            if (!merge) {
                std::memcpy(cursor_transform, instances[i].transform, 48);
                have_cursor = true;
                instances[ic] = instances[i];
                cursor_i = ic++;
                continue;
            }
            std::vector<uint32_t> &dst = mesh_geoms[pmeshes[cp].mesh];
            const std::vector<uint32_t> &src = mesh_geoms[pmeshes[p].mesh];
            dst.insert(dst.end(), src.begin(), src.end());
            offsets[cp].insert(offsets[cp].end(), offsets[p].begin(), offsets[p].end());
        }
        const size_t merged = instances.size() - ic;
        instances.resize(ic);
        if (!merged) return 0;
        // geometries again mesh by mesh (a mesh's geometries are a contiguous range of the geometry table)
        std::vector<RptrGeometryDesc> ng;
        std::vector<std::vector<uint64_t>> np, nn;
        for (size_t m = 0; m < meshes.size(); ++m) {
            meshes[m].first_geometry = (uint32_t)ng.size();
            meshes[m].num_geometries = (uint32_t)mesh_geoms[m].size();
            for (uint32_t g : mesh_geoms[m]) { // (a geometry listed by two meshes after a merge is stored twice)
                ng.push_back(geometries[g]);
                np.push_back(qpos[g]);
                nn.push_back(qnu[g]);
            }
        }
        geometries.swap(ng);
        qpos.swap(np);
        qnu.swap(nn);
        fix_pointers();
        return merged;
    }

    // Scene::deduplicate + garbage_collect (librender/scene.cpp:142-148 and below; `--deduplicate-scene`): meshes with the same content
    // become one mesh, equal materials one material, equal textures one texture; what nothing refers to any more is dropped. Instances keep
    // their order and their parameterized meshes (ray-query ids and the order of the emitters do not change): the image is the same.
    struct DedupStats {
        size_t meshes = 0, materials = 0, textures = 0;
    };
    DedupStats deduplicate() {
        DedupStats st;
        // textures by content
        std::vector<uint32_t> tex_map(textures.size());
        for (size_t i = 0; i < textures.size(); ++i) {
            tex_map[i] = (uint32_t)i;
            for (size_t j = 0; j < i; ++j)
                if (tex_map[j] == j && textures[j].width == textures[i].width && textures[j].height == textures[i].height && textures[j].srgb == textures[i].srgb &&
                    textures[j].mip_levels == textures[i].mip_levels && texels[j] == texels[i]) {
                    tex_map[i] = (uint32_t)j;
                    break;
                }
        }
        for (RptrBaseMaterial &m : materials) for_each_texture_ref(m, [&](uint32_t &id) { if (id < tex_map.size()) id = tex_map[id]; });
        // materials by content
        std::vector<uint32_t> mat_map(materials.size());
        for (size_t i = 0; i < materials.size(); ++i) {
            mat_map[i] = (uint32_t)i;
            for (size_t j = 0; j < i; ++j)
                if (mat_map[j] == j && std::memcmp(&materials[j], &materials[i], sizeof(RptrBaseMaterial)) == 0) {
                    mat_map[i] = (uint32_t)j;
                    break;
                }
        }
        // a parameterized mesh addresses materials as offset + per-triangle id: the offset can only move when the whole range it can reach
        // maps by the same shift, which is the case for ranges without duplicates; otherwise the per-triangle ids are rewritten
        for (size_t p = 0; p < pmeshes.size(); ++p) {
            const RptrMeshDesc &mesh = meshes[pmeshes[p].mesh];
            size_t at = 0;
            for (uint32_t j = 0; j < mesh.num_geometries && j < offsets[p].size(); ++j) {
                const uint32_t nt = geometries[mesh.first_geometry + j].num_tris;
                const int32_t off = offsets[p][j];
                if (tri_ids[p].empty()) {
                    if (off >= 0 && (size_t)off < mat_map.size()) offsets[p][j] = (int32_t)mat_map[(size_t)off];
                } else {
                    // new offset: the smallest mapped id of the range; ids become distances to it (they stay bytes: checked)
                    uint32_t lo = 0xFFFFFFFFu, hi = 0;
                    for (uint32_t t = 0; t < nt; ++t) {
                        const uint32_t m = mat_map[(size_t)off + tri_ids[p][at + t]];
                        lo = std::min(lo, m);
                        hi = std::max(hi, m);
                    }
                    if (nt && hi - lo < 256u) {
                        for (uint32_t t = 0; t < nt; ++t) tri_ids[p][at + t] = (uint8_t)(mat_map[(size_t)off + tri_ids[p][at + t]] - lo);
                        offsets[p][j] = (int32_t)lo;
                    }
                }
                at += nt;
            }
        }
        // meshes by content (geometry streams, quantisation, flags)
        auto same_mesh = [&](const RptrMeshDesc &a, const RptrMeshDesc &b) {
            if (a.num_geometries != b.num_geometries || a.dynamic != b.dynamic) return false;
            for (uint32_t j = 0; j < a.num_geometries; ++j) {
                const size_t ga = a.first_geometry + j, gb = b.first_geometry + j;
                const RptrGeometryDesc &x = geometries[ga], &y = geometries[gb];
                if (x.num_tris != y.num_tris || x.has_normals != y.has_normals || x.has_uvs != y.has_uvs || std::memcmp(x.quantized_scaling, y.quantized_scaling, 12) ||
                    std::memcmp(x.quantized_offset, y.quantized_offset, 12) || qpos[ga] != qpos[gb] || qnu[ga] != qnu[gb])
                    return false;
            }
            return true;
        };
        std::vector<uint32_t> mesh_map(meshes.size());
        for (size_t i = 0; i < meshes.size(); ++i) {
            mesh_map[i] = (uint32_t)i;
            for (size_t j = 0; j < i; ++j)
                if (mesh_map[j] == j && same_mesh(meshes[j], meshes[i])) {
                    mesh_map[i] = (uint32_t)j;
                    break;
                }
        }
        for (RptrParameterizedMeshDesc &pm : pmeshes) pm.mesh = mesh_map[pm.mesh];
        // ---- garbage collection: meshes (with their geometries), materials, textures nothing refers to
        std::vector<char> mesh_used(meshes.size(), 0), mat_used(materials.size(), 0), tex_used(textures.size(), 0);
        for (const RptrParameterizedMeshDesc &pm : pmeshes) mesh_used[pm.mesh] = 1;
        for (size_t p = 0; p < pmeshes.size(); ++p) {
            const RptrMeshDesc &mesh = meshes[pmeshes[p].mesh];
            size_t at = 0;
            for (uint32_t j = 0; j < mesh.num_geometries && j < offsets[p].size(); ++j) {
                const uint32_t nt = geometries[mesh.first_geometry + j].num_tris;
                if (tri_ids[p].empty()) {
                    if ((size_t)offsets[p][j] < mat_used.size()) mat_used[(size_t)offsets[p][j]] = 1;
                } else
                    for (uint32_t t = 0; t < nt; ++t) {
                        const size_t m = (size_t)offsets[p][j] + tri_ids[p][at + t];
                        if (m < mat_used.size()) mat_used[m] = 1;
                    }
                at += nt;
            }
        }
        // (materials are addressed as offset + id: holes inside a used range must survive, so only a used / unused PREFIX structure is
        // compacted: a material is dropped when it is unused AND every range that spans it is rewritten -- kept simple: unused materials
        // at the tail are dropped)
        size_t keep_mats = materials.size();
        while (keep_mats > 0 && !mat_used[keep_mats - 1]) --keep_mats;
        st.materials = materials.size() - keep_mats;
        materials.resize(keep_mats);
        for (const RptrBaseMaterial &m : materials) {
            RptrBaseMaterial c = m;
            for_each_texture_ref(c, [&](uint32_t &id) { if (id < tex_used.size()) tex_used[id] = 1; });
        }
        std::vector<uint32_t> tex_new(textures.size(), 0);
        {
            std::vector<std::vector<uint8_t>> nt;
            std::vector<RptrTextureDesc> nd;
            for (size_t i = 0; i < textures.size(); ++i)
                if (tex_used[i]) {
                    tex_new[i] = (uint32_t)nd.size();
                    nt.push_back(std::move(texels[i]));
                    nd.push_back(textures[i]);
                } else
                    ++st.textures;
            texels.swap(nt);
            textures.swap(nd);
        }
        for (RptrBaseMaterial &m : materials) for_each_texture_ref(m, [&](uint32_t &id) { if (id < tex_new.size()) id = tex_new[id]; });
        std::vector<uint32_t> mesh_new(meshes.size(), 0);
        {
            std::vector<RptrMeshDesc> nm;
            std::vector<RptrGeometryDesc> ng;
            std::vector<std::vector<uint64_t>> np, nn;
            for (size_t i = 0; i < meshes.size(); ++i) {
                if (!mesh_used[i]) {
                    ++st.meshes;
                    continue;
                }
                RptrMeshDesc m = meshes[i];
                mesh_new[i] = (uint32_t)nm.size();
                const uint32_t first = (uint32_t)ng.size();
                for (uint32_t j = 0; j < m.num_geometries; ++j) {
                    ng.push_back(geometries[m.first_geometry + j]);
                    np.push_back(std::move(qpos[m.first_geometry + j]));
                    nn.push_back(std::move(qnu[m.first_geometry + j]));
                }
                m.first_geometry = first;
                nm.push_back(m);
            }
            meshes.swap(nm);
            geometries.swap(ng);
            qpos.swap(np);
            qnu.swap(nn);
        }
        for (RptrParameterizedMeshDesc &pm : pmeshes) pm.mesh = mesh_new[pm.mesh];
        fix_pointers();
        return st;
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
