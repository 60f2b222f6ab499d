// host_scene.inl -- rptr_hip_set_scene (upload, device tables, shading records, per-context scene copies), vertex updates, refit and device rebuild
// Part of the ONE translation unit rptr_hip.hip (included there, in this order: host_state.h, host_bvh.inl, host_scene.inl,
// host_frame.inl, host_access.inl, host_comm.h): the host runtime split along its seams; no symbol changed.
// ---- set_scene, step by step (each returns RPTR_OK or the error it reported through fail())
// what the reference host rejects or this build does not cover yet; sets uses_textures / uses_alpha
static int scene_validate(rptr_hip *h, const RptrSceneDesc *s) {
    {
        const std::string bad = validate_scene_tables(s);
        if (!bad.empty()) return fail(h, RPTR_E_INVALID, "%s", bad.c_str());
    }
    if (s->num_textures && !s->textures) return fail(h, RPTR_E_INVALID, "num_textures = %u but textures is NULL", s->num_textures);
    for (uint32_t t = 0; t < s->num_textures; ++t) {
        if (!s->textures[t].rgba8 || s->textures[t].width == 0 || s->textures[t].height == 0 || s->textures[t].width > 16384 || s->textures[t].height > 16384)
            return fail(h, RPTR_E_INVALID, "texture %u: bad size or NULL data", t);
        if (s->textures[t].mip_levels > 1u) { // at most the full chain down to 1 x 1
            uint32_t full = 1;
            for (uint32_t w = s->textures[t].width, hh = s->textures[t].height; w > 1 || hh > 1; w = std::max(1u, w / 2), hh = std::max(1u, hh / 2)) ++full;
            if (s->textures[t].mip_levels > full)
                return fail(h, RPTR_E_INVALID, "texture %u: %u mip levels, a %u x %u texture has at most %u", t, s->textures[t].mip_levels, s->textures[t].width,
                            s->textures[t].height, full);
        }
    }
    h->uses_textures = false;
    h->uses_alpha = false;
    h->tail_adaptive = 1 << 30; // the first frame of a scene shows the queue lengths of every bounce
    for (uint32_t m = 0; m < s->num_materials; ++m) {
        const RptrBaseMaterial &mat = s->materials[m];
        if (mat.normal_map != -1) h->uses_textures = true;
        if (mat.normal_map != -1 && (mat.normal_map < 0 || (uint32_t)mat.normal_map >= s->num_textures))
            return fail(h, RPTR_E_INVALID, "material %u: normal_map %d is not a texture of this scene (%u textures)", m, mat.normal_map, s->num_textures);
        if ((mat.flags & RPTR_BASE_MATERIAL_NOALPHA) == 0) h->uses_alpha = true; // alpha test of hit candidates (kernels.h ALPHA)
        const float vals[5] = {mat.base_color[0], mat.roughness, mat.specular, mat.metallic, mat.ior};
        for (float v : vals) {
            uint32_t u;
            memcpy(&u, &v, 4);
            if (u & RPTR_TEXTURED_PARAM_MASK) h->uses_textures = true;
            if ((u & RPTR_TEXTURED_PARAM_MASK) && RPTR_TEXTURE_ID(u) >= s->num_textures)
                return fail(h, RPTR_E_INVALID, "material %u: textured parameter refers to texture %u of %u", m, RPTR_TEXTURE_ID(u), s->num_textures);
        }
    }
    return RPTR_OK;
}
// textures (RGBA8, mip levels back to back) + the sRGB decode table
static int scene_upload_textures(rptr_hip *h, const RptrSceneDesc *s, RpTexture *&d_textures, float *&d_srgb_lut) {
    int rc;
    d_textures = nullptr;
    d_srgb_lut = nullptr;
    // paths through a scene with textures carry their texture footprint (kernels.h TEX; the tail kernel's textured instantiation also serves
    // alpha-tested scenes)
    if ((h->uses_textures || h->uses_alpha) && h->path_capacity)
        for (FrameCtx &c : h->ctx)
            if (!c.ps.footprint && (rc = dev_alloc(h, &c.ps.footprint, h->path_capacity, nullptr))) return rc;
    {
        std::vector<RpTexture> tex(s->num_textures);
        for (uint32_t t = 0; t < s->num_textures; ++t) {
            const RptrTextureDesc &td = s->textures[t];
            uchar4 *dt = nullptr;
            const uint32_t levels = td.mip_levels > 1u ? td.mip_levels : 1u;
            size_t n = 0; // the levels back to back, level l = max(1, w >> l) x max(1, h >> l) (vulkan/resource_utils.cpp:86-100)
            for (uint32_t l = 0, w = td.width, hh = td.height; l < levels; ++l, w = std::max(1u, w / 2), hh = std::max(1u, hh / 2)) n += (size_t)w * hh;
            if ((rc = dev_alloc(h, &dt, n, &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(dt, td.rgba8, n * 4, hipMemcpyHostToDevice));
            tex[t].texels = dt;
            tex[t].width = (int)td.width;
            tex[t].height = (int)td.height;
            tex[t].srgb = td.srgb ? 1 : 0;
            tex[t].levels = (int)levels;
        }
        if ((rc = dev_alloc(h, &d_textures, std::max<size_t>(1, tex.size()), &h->scene_allocs))) return rc;
        if (!tex.empty()) HIP_TRY(h, hipMemcpy(d_textures, tex.data(), tex.size() * sizeof(RpTexture), hipMemcpyHostToDevice));
        float lut[256]; // IEC 61966-2-1 decode of an 8-bit code (what a VK_FORMAT_*_SRGB fetch returns before filtering)
        for (int i = 0; i < 256; ++i) {
            const float c = float(i) / 255.0f;
            lut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
        }
        if ((rc = dev_alloc(h, &d_srgb_lut, 256, &h->scene_allocs))) return rc;
        HIP_TRY(h, hipMemcpy(d_srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    }
    return RPTR_OK;
}
// the quantised vertex streams, one allocation per stream; dynamic meshes keep full-precision float positions next to them
static int scene_upload_vertex_streams(rptr_hip *h, const RptrSceneDesc *s, std::vector<const uint64_t *> &d_qpos, std::vector<const uint64_t *> &d_qnu) {
    int rc;
    d_qpos.assign(s->num_geometries, nullptr);
    d_qnu.assign(s->num_geometries, nullptr);
    for (uint32_t g = 0; g < s->num_geometries; ++g) {
        const RptrGeometryDesc &gd = s->geometries[g];
        uint64_t *dp = nullptr;
        if ((rc = dev_alloc(h, &dp, (size_t)gd.num_tris * 3, &h->scene_allocs))) return rc;
        if (gd.num_tris) HIP_TRY(h, hipMemcpy(dp, gd.qpos, (size_t)gd.num_tris * 24, hipMemcpyHostToDevice));
        d_qpos[g] = dp;
        if (gd.qnrm_uv && (gd.has_normals || gd.has_uvs)) {
            uint64_t *dn = nullptr;
            if ((rc = dev_alloc(h, &dn, (size_t)gd.num_tris * 3, &h->scene_allocs))) return rc;
            if (gd.num_tris) HIP_TRY(h, hipMemcpy(dn, gd.qnrm_uv, (size_t)gd.num_tris * 24, hipMemcpyHostToDevice));
            d_qnu[g] = dn;
        }
    }
    // ---- dynamic meshes keep full-precision float positions next to the quantised stream
    h->master.dynpos.assign(s->num_geometries, nullptr);
    h->geom_tris.assign(s->num_geometries, 0);
    h->geom_mesh.assign(s->num_geometries, -1);
    h->master.mesh_dirty.assign(s->num_meshes, 0);
    h->master.mesh_dyn.assign(s->num_meshes, nullptr);
    for (uint32_t m = 0; m < s->num_meshes; ++m) {
        const RptrMeshDesc &mesh = s->meshes[m];
        std::vector<const float *> table(mesh.num_geometries, nullptr);
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t gi = mesh.first_geometry + j;
            const RptrGeometryDesc &gd = s->geometries[gi];
            h->geom_tris[gi] = gd.num_tris;
            h->geom_mesh[gi] = (int)m;
            if (!mesh.dynamic) continue;
            std::vector<float> pos((size_t)gd.num_tris * 9);
            for (size_t v = 0; v < (size_t)gd.num_tris * 3; ++v) dequantize_position(gd.qpos[v], gd.quantized_scaling, gd.quantized_offset, &pos[3 * v]);
            float *dp = nullptr;
            if ((rc = dev_alloc(h, &dp, pos.size(), &h->scene_allocs))) return rc;
            if (!pos.empty()) HIP_TRY(h, hipMemcpy(dp, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice));
            h->master.dynpos[gi] = dp;
            table[j] = dp;
        }
        if (mesh.dynamic) {
            const float **dt = nullptr;
            if ((rc = dev_alloc(h, &dt, table.size(), &h->scene_allocs))) return rc;
            if (!table.empty()) HIP_TRY(h, hipMemcpy(dt, table.data(), table.size() * sizeof(float *), hipMemcpyHostToDevice));
            h->master.mesh_dyn[m] = dt;
            h->master.mesh_dirty[m] = 2;
        }
    }
    return RPTR_OK;
}
// geometry records per (parameterized mesh, geometry): instanced_geometry[] (render_vulkan.cpp:2748-2850)
static int scene_geometry_records(rptr_hip *h, const RptrSceneDesc *s, const std::vector<const uint64_t *> &d_qpos, const std::vector<const uint64_t *> &d_qnu,
                                  std::vector<RpGeomRecord> &geoms, std::vector<int> &pmesh_base) {
    int rc;
    geoms.clear();
    pmesh_base.assign(s->num_parameterized_meshes, 0);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        pmesh_base[p] = (int)geoms.size();
        size_t total_tris = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) total_tris += s->geometries[mesh.first_geometry + j].num_tris;
        uint8_t *d_ids = nullptr;
        if (pm.tri_material_ids) {
            if ((rc = dev_alloc(h, &d_ids, total_tris, &h->scene_allocs))) return rc;
            if (total_tris) HIP_TRY(h, hipMemcpy(d_ids, pm.tri_material_ids, total_tris, hipMemcpyHostToDevice));
        }
        size_t prim_offset = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t gi = mesh.first_geometry + j;
            const RptrGeometryDesc &gd = s->geometries[gi];
            RpGeomRecord r;
            memset(&r, 0, sizeof(r));
            r.qpos = d_qpos[gi];
            r.qnrm_uv = d_qnu[gi];
            r.mat_ids = d_ids ? d_ids + prim_offset : nullptr;
            r.dyn_pos = h->master.dynpos[gi];
            memcpy(r.scaling, gd.quantized_scaling, 12);
            memcpy(r.offset, gd.quantized_offset, 12);
            r.material_id = d_ids ? -1 - pm.material_offsets[j] : pm.material_offsets[j];
            r.flags = (gd.has_normals && d_qnu[gi] ? RP_GEOM_HAS_NORMALS : 0u) | (gd.has_uvs && d_qnu[gi] ? RP_GEOM_HAS_UVS : 0u) |
                      (h->master.dynpos[gi] ? RP_GEOM_DYNAMIC : 0u);
            geoms.push_back(r);
            prim_offset += gd.num_tris;
        }
    }
    return RPTR_OK;
}
// the acceleration structure: per-mesh trees (host binned SAH, or the device's PLOC builder for large static sets), top level, encoding
static int scene_build_acceleration_structure(rptr_hip *h, const RptrSceneDesc *s, std::vector<const uint64_t *> &d_qpos, std::vector<RpGeomRecord> &geoms, HostBvh &B) {
    {
        // large static triangle sets are built on the device (csrc/ploc.h) from the vertex streams uploaded above
        std::vector<uint8_t> mat_alpha(s->num_materials, 0);
        for (uint32_t i = 0; i < s->num_materials; ++i) mat_alpha[i] = (s->materials[i].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0 ? 1 : 0;
        DeviceBuildCtx ctx;
        ctx.d_qpos = &d_qpos;
        ctx.geoms = &geoms;
        ctx.min_tris = (size_t)h->opt.v[OPT_DEVICE_BUILD_MIN_TRIS];
        int device_failures = 0;
        std::string device_failure;
        ctx.build = [&](const std::vector<RpBuildSegment> &segs, uint32_t n, DeviceTree &out) {
            const bool ok = device_build_tree(h, segs, n, mat_alpha, out);
            if (!ok) { // the host builder takes over (same scene, seconds instead of a fraction of one): say so, and do not leave the
                       // message behind as the "last error" of a call that succeeds
                ++device_failures;
                device_failure = h->last_error;
                h->last_error.clear();
            }
            return ok;
        };
        const auto t_build = std::chrono::steady_clock::now();
        build_host_bvh(s, B, h->opt, &ctx);
        if (device_failures && h->opt.v[OPT_QUIET] == 0)
            fprintf(stderr, "rptr_hip: note: %d device-side BVH build(s) failed (%s); the host builder built those trees instead\n", device_failures,
                    device_failure.c_str());
        h->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
        h->bvh_device_built = B.device_built;
        h->bvh_device_ms = B.device_ms;
    }
    {
        const int capacity = RP_LDS_STACK + RPTR_BVH_STACK_DEPTH;
        if (B.stack_need > capacity)
            return fail(h, RPTR_E_UNSUPPORTED, "the acceleration structure of this scene needs a traversal stack of %d entries (limit %d)",
                        B.stack_need, capacity);
    }
    return RPTR_OK;
}
// refit tables: the top level by height (children before parents); the depth levels of every dynamic mesh's tree
static int scene_refit_tables(rptr_hip *h, std::vector<uint32_t> &refit_list, std::vector<uint32_t> &h_blas_list, std::vector<std::array<uint2, RP_REFIT_LEVELS>> &h_levels) {
    refit_list.clear();
    h_blas_list.assign(h->h_nodes.size(), 0u);
    h_levels.assign(h->meshes.size(), std::array<uint2, RP_REFIT_LEVELS>());
    // (the bottom-level trees of dynamic meshes are refitted bottom-up with arrival counters, lbvh.h rp_k_refit_up: per node its parent and the
    // number of its inner children)
    h->refit_levels_tlas.clear();
    h->has_dynamic = false;
    for (const MeshRt &mr : h->meshes) h->has_dynamic = h->has_dynamic || mr.dynamic;
    // depth levels of every dynamic mesh's tree (slot RP_REFIT_LEVELS - 1 - depth: ascending slot = deepest first)
    {
        const size_t nn = h->h_nodes.size();
        std::vector<int> height(nn, -1);
        std::vector<std::vector<uint32_t>> tlas_levels;
        // iterative post-order: height = 1 + max(height of inner children), 0 for nodes with leaf children only
        std::vector<std::pair<int, int>> st{{0, 0}};
        while (!st.empty()) {
            auto [n, phase] = st.back();
            st.pop_back();
            const RptrBvh4Node &nd = h->h_nodes[n];
            if (phase == 0) {
                st.push_back({n, 1});
                for (int k = 0; k < 4; ++k)
                    if (nd.child[k] >= 0) st.push_back({nd.child[k], 0});
            } else {
                int hgt = 0;
                for (int k = 0; k < 4; ++k)
                    if (nd.child[k] >= 0) hgt = std::max(hgt, height[nd.child[k]] + 1);
                height[n] = hgt;
                if ((size_t)hgt >= tlas_levels.size()) tlas_levels.resize(hgt + 1);
                tlas_levels[hgt].push_back((uint32_t)n | 0x80000000u);
            }
        }
        for (auto &lv : tlas_levels) {
            h->refit_levels_tlas.push_back({(uint32_t)refit_list.size(), (uint32_t)(refit_list.size() + lv.size())});
            refit_list.insert(refit_list.end(), lv.begin(), lv.end());
        }
        for (size_t m = 0; m < h->meshes.size(); ++m) {
            const MeshRt &mr = h->meshes[m];
            for (auto &l : h_levels[m]) l = make_uint2((uint32_t)mr.node_base, (uint32_t)mr.node_base);
            if (!mr.dynamic) continue;
            std::vector<std::vector<uint32_t>> by_depth;
            std::vector<std::pair<int, int>> bfs{{h->mesh_root[m], 0}};
            for (size_t at = 0; at < bfs.size(); ++at) {
                const auto [n, d] = bfs[at];
                if ((size_t)d >= by_depth.size()) by_depth.resize((size_t)d + 1);
                by_depth[(size_t)d].push_back((uint32_t)n);
                for (int k = 0; k < 4; ++k)
                    if (h->h_nodes[(size_t)n].child[k] >= 0) bfs.push_back({h->h_nodes[(size_t)n].child[k], d + 1});
            }
            if (by_depth.size() > RP_REFIT_LEVELS) return fail(h, RPTR_E_UNSUPPORTED, "mesh %zu: a tree of %zu levels (limit %d)", m, by_depth.size(), RP_REFIT_LEVELS);
            uint32_t at = (uint32_t)mr.node_base;
            for (int slot = 0; slot < RP_REFIT_LEVELS; ++slot) {
                const int d = RP_REFIT_LEVELS - 1 - slot;
                const uint32_t cnt = (size_t)d < by_depth.size() ? (uint32_t)by_depth[(size_t)d].size() : 0u;
                h_levels[m][(size_t)slot] = make_uint2(at, at + cnt);
                for (uint32_t k = 0; k < cnt; ++k) h_blas_list[at + k] = by_depth[(size_t)d][k];
                at += cnt;
            }
        }
    }
    return RPTR_OK;
}
// the traversal's scheduling thresholds and pool size for this scene's trees (RpScene.node_min / refill_min / fetch_max)
static void scene_traversal_preset(rptr_hip *h) {
    // Scheduling thresholds of the traversal (dtraverse.h): a wave refills its idle lanes together once `refill_min` of them have finished,
    // and leaves a node phase for a leaf phase once fewer than `node_min` lanes are at inner nodes. The defaults (48 / 10) were tuned on the
    // height field; in a dense soup of overlapping primitive boxes -- the forest: 26 node visits per ray, node-phase lane utilisation 0.55
    // instead of 0.67, a quarter of the lane slots waiting for a refill -- 32 / 16 are 5 % faster (C4 5.89 -> 5.61 ms) and 1.4 % slower on the
    // height field (profiles/r03_notes.md section 7). The choice follows the tree: the surface-area cost of its largest bottom-level tree
    // (sum of the inner children's box areas over the root's: 11 for the height field, 92 for the flattened forest). RPTR_TRAVERSE_PRESET=
    // "node_min,refill_min" overrides (0,0 = the compile-time defaults).
    {
        double best_cost = 0.0;
        size_t best_tris = 0;
        auto half_area = [&](size_t n) {
            const std::array<float, 6> &b = h->h_node_box[n];
            const double dx = std::max(0.0f, b[3] - b[0]), dy = std::max(0.0f, b[4] - b[1]), dz = std::max(0.0f, b[5] - b[2]);
            return dx * dy + dy * dz + dz * dx;
        };
        for (size_t m = 0; m < h->meshes.size(); ++m) {
            const MeshRt &mr = h->meshes[m];
            // (a mesh without a tree of its own is part of the flattened tree, which lies first: counted once, for mesh 0)
            const size_t root = (size_t)h->mesh_root[m], count = (size_t)(mr.node_count > 0 ? mr.node_count : (m == 0 ? (int)h->flat_nodes : 0));
            const size_t tris_m = mr.tri_count > 0 ? (size_t)mr.tri_count : (m == 0 ? h->flat_tris : 0);
            if (!count || tris_m < best_tris || root >= h->h_nodes.size()) continue;
            const double a0 = half_area(root);
            if (!(a0 > 0.0)) continue;
            double sum = 0.0;
            for (size_t n = root; n < std::min(root + count, h->h_nodes.size()); ++n)
                for (int k = 0; k < 4; ++k)
                    if (h->h_nodes[n].child[k] >= 0) sum += half_area((size_t)h->h_nodes[n].child[k]);
            best_cost = sum / a0;
            best_tris = tris_m;
        }
        // ... times the same measure of the top level (all child boxes of its nodes, instance boxes included, over the scene's box: 1 for a
        // single instance, ~ 6 for the forest's 1001 overlapping instances: the two-level forest gains the same 5 %, 8.38 -> 7.95 ms)
        double tlas_cost = 1.0;
        if (h->num_tlas_nodes > 0 && h->num_tlas_insts > 1) {
            const double a0 = half_area(0);
            double sum = 0.0;
            for (int n = 0; n < h->num_tlas_nodes; ++n) {
                const RptrBvh4Node &nd = h->h_nodes[(size_t)n];
                for (int k = 0; k < 4; ++k) {
                    if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                    double d[3];
                    for (int a = 0; a < 3; ++a) d[a] = std::max(0.0, (double)((int)nd.qhi[a][k] - (int)nd.qlo[a][k])) * std::ldexp(1.0, (int)nd.exp[a] - 127);
                    sum += d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
                }
            }
            if (a0 > 0.0) tlas_cost = std::max(1.0, sum / a0);
        }
        best_cost *= tlas_cost;
        h->bvh_area_cost = best_cost;
        int node_min = 0, refill_min = 0;
        if (best_cost >= 24.0) { // (height fields: 10-12; the small forests of the tests: 29-35; C4: 90 flattened, 250-280 two-level)
            node_min = 16;
            refill_min = 32;
        }
        if (h->opt.v[OPT_TRAVERSE_NODE_MIN] >= 0) { // options "traverse_node_min" / "traverse_refill_min" (0, 0: the compile-time defaults)
            node_min = (int)h->opt.v[OPT_TRAVERSE_NODE_MIN];
            refill_min = (int)std::max(0ll, h->opt.v[OPT_TRAVERSE_REFILL_MIN]);
        }
        h->master.dscene.node_min = std::max(0, std::min(64, node_min));
        h->master.dscene.refill_min = std::max(0, std::min(64, refill_min));
        h->master.dscene.lds_top = h->opt.v[OPT_LDS_TOP] != 0 ? 1 : 0;
        // ... and the size of a wave's pool of queue entries (dtraverse.h RP_FETCH: 256, four tiles of the first queue): 384 for the trees
        // of the default preset -- one frame at a time 1.82 -> 1.76 ms, two in flight 1.45 -> 1.38 on C2, pipelined unchanged --, 256 for dense
        // ones (the forest loses 4 % with 384; profiles/r05_notes.md section 19)
        h->master.dscene.fetch_max = h->opt.v[OPT_TRAVERSE_FETCH] > 0 ? (int)std::max(64ll, h->opt.v[OPT_TRAVERSE_FETCH] / 64 * 64) : (best_cost >= 24.0 ? 0 : 384);
    }
}
int rptr_hip_set_scene(rptr_hip_t *h, const RptrSceneDesc *s) {
    if (!h || !s) return fail(h, RPTR_E_INVALID, "NULL argument");
    HIP_TRY(h, hipSetDevice(h->device));
    {
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (void *p : h->scene_allocs) {
        (void)hipFree(p);
    }
    h->scene_allocs.clear();
    h->bytes_scene = 0;
    h->bytes_allocated = h->bytes_frame;
    h->have_scene = false;
    int rc;
    if ((rc = scene_validate(h, s))) return rc;
    // ---- uploads: textures, vertex streams, geometry records
    RpTexture *d_textures = nullptr;
    float *d_srgb_lut = nullptr;
    if ((rc = scene_upload_textures(h, s, d_textures, d_srgb_lut))) return rc;
    std::vector<const uint64_t *> d_qpos, d_qnu;
    if ((rc = scene_upload_vertex_streams(h, s, d_qpos, d_qnu))) return rc;
    std::vector<RpGeomRecord> geoms;
    std::vector<int> pmesh_base;
    if ((rc = scene_geometry_records(h, s, d_qpos, d_qnu, geoms, pmesh_base))) return rc;
    // ---- acceleration structure
    HostBvh B;
    if ((rc = scene_build_acceleration_structure(h, s, d_qpos, geoms, B))) return rc;
    h->h_nodes = std::move(B.nodes);
    h->h_node_box = std::move(B.node_box);
    h->h_tris = std::move(B.tris);
    h->h_insts = std::move(B.insts);
    h->num_tlas_insts = B.num_tlas_insts;
    h->meshes = std::move(B.meshes);
    h->mesh_root = std::move(B.mesh_root);
    h->num_tlas_nodes = B.num_tlas_nodes;
    h->flat_tris = B.flat_tris;
    h->flat_nodes = B.flat_nodes;
    memcpy(h->scene_lo, B.scene_lo, 12);
    memcpy(h->scene_hi, B.scene_hi, 12);
    // ---- refit tables
    std::vector<uint32_t> refit_list, h_blas_list;
    std::vector<std::array<uint2, RP_REFIT_LEVELS>> h_levels;
    if ((rc = scene_refit_tables(h, refit_list, h_blas_list, h_levels))) return rc;
    h->rebuild_epoch.assign(h->meshes.size(), 0);
    h->bvh_credit = 0;
    h->rebuild_cursor = 0;
    // ---- upload
    RptrBvh4Node *d_nodes = nullptr;
    RptrBvhTri *d_tris = nullptr;
    RptrBvhInstance *d_insts = nullptr;
    RpGeomRecord *d_geoms = nullptr;
    RptrBaseMaterial *d_mats = nullptr;
    RptrTriLightData *d_lights = nullptr;
    if ((rc = dev_alloc(h, &d_nodes, h->h_nodes.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_tris, h->h_tris.size() + 2, &h->scene_allocs))) return rc; // +2: a leaf is fetched as whole pairs
    if ((rc = dev_alloc(h, &d_insts, h->h_insts.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_geoms, geoms.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_mats, s->num_materials, &h->scene_allocs))) return rc;
    // light buffer padded with one zeroed bin (+1): sample_tri_lights may read light_id == bin_end
    const size_t light_cap = (size_t)s->num_lights + RPTR_BINNED_LIGHTS_BIN_MAX_SIZE + 1;
    if ((rc = dev_alloc(h, &d_lights, light_cap, &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &h->d_refit_list, refit_list.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &h->master.inst_box, (size_t)6 * h->h_insts.size(), &h->scene_allocs))) return rc;
    h->master.tri_box = nullptr;
    if (h->has_dynamic && (rc = dev_alloc(h, &h->master.tri_box, (size_t)6 * h->h_tris.size(), &h->scene_allocs))) return rc;
    if (!refit_list.empty()) HIP_TRY(h, hipMemcpy(h->d_refit_list, refit_list.data(), refit_list.size() * 4, hipMemcpyHostToDevice));
    {
        // the instance bounds and the (small) top-level levels of a refit share one launch (kernels.h rp_k_refit_top): every level is one
        // more dependent launch otherwise, and an animated frame pays for them whatever its size
        const uint32_t small = 4096;
        std::vector<uint2> lv;
        for (auto &l : h->refit_levels_tlas) lv.push_back(make_uint2(l[0], l[1]));
        h->refit_top_all = (size_t)h->num_tlas_insts <= 4 * small; // (only the records the top level refers to have bounds: a flattened tree's triangles name the others)
        for (auto &l : h->refit_levels_tlas) h->refit_top_all = h->refit_top_all && l[1] - l[0] <= small;
        h->d_refit_levels = nullptr;
        if (!lv.empty()) {
            if ((rc = dev_alloc(h, &h->d_refit_levels, lv.size(), &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(h->d_refit_levels, lv.data(), lv.size() * sizeof(uint2), hipMemcpyHostToDevice));
        }
    }
    // level tables + node lists of the dynamic meshes + per-mesh node counts (one set per scene copy: copies are rebuilt independently)
    auto make_refit_tables = [&](SceneCopy &sc) -> int {
        int rc2;
        sc.device_built.assign(h->meshes.size(), 0);
        sc.built_epoch.assign(h->meshes.size(), 0);
        sc.scratch = RpLbvhScratch();
        sc.blas_list = nullptr;
        sc.blas_levels = nullptr;
        sc.mesh_count = nullptr;
        sc.host_levels = h_levels;
        sc.levels_known.assign(h->meshes.size(), 1);
        release_scene_copy_host(sc);
        sc.pinned_levels.assign(h->meshes.size(), nullptr);
        sc.ev_levels.assign(h->meshes.size(), nullptr);
        if (!h->has_dynamic) return RPTR_OK;
        if ((rc2 = dev_alloc(h, &sc.blas_list, h->h_nodes.size(), &h->scene_allocs))) return rc2;
        if ((rc2 = dev_alloc(h, &sc.blas_levels, h->meshes.size() * RP_REFIT_LEVELS, &h->scene_allocs))) return rc2;
        if ((rc2 = dev_alloc(h, &sc.mesh_count, h->meshes.size(), &h->scene_allocs))) return rc2;
        HIP_TRY(h, hipMemcpy(sc.blas_list, h_blas_list.data(), h_blas_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(sc.blas_levels, h_levels.data(), h_levels.size() * sizeof(h_levels[0]), hipMemcpyHostToDevice));
        std::vector<int> counts;
        for (const MeshRt &mr : h->meshes) counts.push_back(mr.node_count);
        HIP_TRY(h, hipMemcpy(sc.mesh_count, counts.data(), counts.size() * sizeof(int), hipMemcpyHostToDevice));
        for (size_t m = 0; m < h->meshes.size(); ++m)
            if (h->meshes[m].dynamic) {
                if (hipHostMalloc((void **)&sc.pinned_levels[m], RP_REFIT_LEVELS * sizeof(uint2), hipHostMallocDefault) != hipSuccess)
                    return fail(h, RPTR_E_NOMEM, "hipHostMalloc failed");
                HIP_TRY(h, hipEventCreateWithFlags(&sc.ev_levels[m], hipEventDisableTiming));
            }
        return RPTR_OK;
    };
    if ((rc = make_refit_tables(h->master))) return rc;
    h->host_bvh_stale = false;
    h->master_refit_pending = false;
    if ((rc = dev_alloc(h, &h->master.node_box, (size_t)6 * h->h_nodes.size(), &h->scene_allocs))) return rc;
    HIP_TRY(h, hipMemcpy(h->master.node_box, h->h_node_box.data(), h->h_node_box.size() * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_nodes, h->h_nodes.data(), h->h_nodes.size() * sizeof(RptrBvh4Node), hipMemcpyHostToDevice));
    if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(d_tris, h->h_tris.data(), h->h_tris.size() * sizeof(RptrBvhTri), hipMemcpyHostToDevice));
    if (!h->h_insts.empty())
        HIP_TRY(h, hipMemcpy(d_insts, h->h_insts.data(), h->h_insts.size() * sizeof(RptrBvhInstance), hipMemcpyHostToDevice));
    if (!geoms.empty()) HIP_TRY(h, hipMemcpy(d_geoms, geoms.data(), geoms.size() * sizeof(RpGeomRecord), hipMemcpyHostToDevice));
    if (s->num_materials) HIP_TRY(h, hipMemcpy(d_mats, s->materials, s->num_materials * sizeof(RptrBaseMaterial), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemset(d_lights, 0, light_cap * sizeof(RptrTriLightData)));
    if (s->num_lights) HIP_TRY(h, hipMemcpy(d_lights, s->lights, s->num_lights * sizeof(RptrTriLightData), hipMemcpyHostToDevice));
    h->master.nodes = d_nodes;
    h->master.tris = d_tris;
    h->master.version = h->refit_version;
    h->master.dscene.nodes = d_nodes;
    h->master.dscene.tris = d_tris;
    h->master.dscene.insts = d_insts;
    h->master.dscene.geoms = d_geoms;
    h->master.dscene.materials = d_mats;
    h->master.dscene.lights = d_lights;
    h->master.dscene.num_lights = (int)s->num_lights;
    h->master.dscene.num_materials = (int)s->num_materials;
    h->master.dscene.num_nodes = (uint32_t)h->h_nodes.size();
    h->master.dscene.flat_id_bias = B.flat_id_bias > 0 ? B.flat_id_bias : 1;
    h->master.dscene.single_instance = (h->num_tlas_insts == 1 && h->opt.v[OPT_SINGLE_INSTANCE] != 0) ? 1 : 0;
    h->master.dscene.num_textures = (int)s->num_textures;
    h->master.dscene.textures = d_textures;
    h->master.dscene.srgb_lut = d_srgb_lut;
    scene_traversal_preset(h);
    // ---- one shading record per BVH triangle (dshade.h RpShadeTri), made on the device from what was just uploaded: per mesh with the
    // geometry records of the first parameterized mesh that uses it, or -- a flattened scene -- per triangle through the instance it names
    {
        RpShadeTri *d_shade = nullptr;
        if ((rc = dev_alloc(h, &d_shade, h->h_tris.size() + 1, &h->scene_allocs))) return rc;
        h->master.shade = d_shade;
        h->master.dscene.shade = d_shade;
        h->mesh_geometry_base.assign(s->num_meshes, -1);
        for (uint32_t p = s->num_parameterized_meshes; p-- > 0;) h->mesh_geometry_base[s->parameterized_meshes[p].mesh] = pmesh_base[p];
        if ((rc = build_shade_records(h, h->master, -1, h->stream))) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->num_lights = (int)s->num_lights;
    h->num_materials = (int)s->num_materials;
    // ---- dynamic scene + frames in flight: every frame context gets its own set of what a refit rewrites
    for (SceneCopy &sc : h->ctx_scene) release_scene_copy_host(sc);
    h->ctx_scene.clear();
    if (h->has_dynamic && h->ctx.size() > 1) {
        h->ctx_scene.resize(h->ctx.size());
        for (SceneCopy &sc : h->ctx_scene) {
            sc.dscene = h->master.dscene;
            sc.mesh_dirty.assign(s->num_meshes, 0);
            sc.dynpos.assign(s->num_geometries, nullptr);
            sc.mesh_dyn.assign(s->num_meshes, nullptr);
            if ((rc = dev_alloc(h, &sc.nodes, h->h_nodes.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.tris, h->h_tris.size() + 2, &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.shade, h->h_tris.size() + 1, &h->scene_allocs))) return rc;
            if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(sc.shade, h->master.shade, h->h_tris.size() * sizeof(RpShadeTri), hipMemcpyDeviceToDevice));
            if ((rc = dev_alloc(h, &sc.node_box, (size_t)6 * h->h_nodes.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.tri_box, (size_t)6 * h->h_tris.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.inst_box, (size_t)6 * h->h_insts.size(), &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(sc.nodes, d_nodes, h->h_nodes.size() * sizeof(RptrBvh4Node), hipMemcpyDeviceToDevice));
            if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(sc.tris, d_tris, h->h_tris.size() * sizeof(RptrBvhTri), hipMemcpyDeviceToDevice));
            HIP_TRY(h, hipMemcpy(sc.node_box, h->master.node_box, h->h_node_box.size() * 24, hipMemcpyDeviceToDevice));
            std::vector<RpGeomRecord> cgeoms = geoms; // same records, pointing at this copy's float positions
            for (uint32_t m = 0; m < s->num_meshes; ++m) {
                const RptrMeshDesc &mesh = s->meshes[m];
                if (!mesh.dynamic) continue;
                std::vector<const float *> table(mesh.num_geometries, nullptr);
                for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                    const uint32_t gi = mesh.first_geometry + j;
                    const size_t nfl = (size_t)s->geometries[gi].num_tris * 9;
                    float *dp = nullptr;
                    if ((rc = dev_alloc(h, &dp, nfl, &h->scene_allocs))) return rc;
                    if (nfl) HIP_TRY(h, hipMemcpy(dp, h->master.dynpos[gi], nfl * sizeof(float), hipMemcpyDeviceToDevice));
                    sc.dynpos[gi] = dp;
                    table[j] = dp;
                }
                const float **dt = nullptr;
                if ((rc = dev_alloc(h, &dt, table.size(), &h->scene_allocs))) return rc;
                if (!table.empty()) HIP_TRY(h, hipMemcpy(dt, table.data(), table.size() * sizeof(float *), hipMemcpyHostToDevice));
                sc.mesh_dyn[m] = dt;
                sc.mesh_dirty[m] = 2;
            }
            for (RpGeomRecord &r : cgeoms)
                if (r.dyn_pos)
                    for (uint32_t gi = 0; gi < s->num_geometries; ++gi)
                        if (r.dyn_pos == h->master.dynpos[gi]) {
                            r.dyn_pos = sc.dynpos[gi];
                            break;
                        }
            RpGeomRecord *cg = nullptr;
            if ((rc = dev_alloc(h, &cg, cgeoms.size(), &h->scene_allocs))) return rc;
            if (!cgeoms.empty()) HIP_TRY(h, hipMemcpy(cg, cgeoms.data(), cgeoms.size() * sizeof(RpGeomRecord), hipMemcpyHostToDevice));
            sc.dscene.nodes = sc.nodes;
            sc.dscene.tris = sc.tris;
            sc.dscene.shade = sc.shade;
            sc.dscene.geoms = cg;
            sc.version = h->refit_version;
            if ((rc = make_refit_tables(sc))) return rc;
        }
    }
    h->have_scene = true;
    // a new scene restarts accumulation (Shell::set_scene -> reset, libapp/shell.cpp:96-126)
    h->frame_offset += h->frame_id;
    h->frame_id = 0;
    return RPTR_OK;
}

static int update_vertices_common(rptr_hip_t *h, uint32_t geometry, const float *xyz, uint32_t num_vertices, bool device_src) {
    if (!h || !xyz) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "update_vertices before set_scene");
    if (h->ctx_scene.empty()) { // frames in flight read the master vertex buffer and tree
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    if (geometry >= h->master.dynpos.size() || !h->master.dynpos[geometry])
        return fail(h, RPTR_E_INVALID, "geometry %u does not belong to a dynamic mesh (RptrMeshDesc.dynamic)", geometry);
    if (num_vertices != 3u * h->geom_tris[geometry])
        return fail(h, RPTR_E_INVALID, "geometry %u has %u unrolled vertices, got %u", geometry, 3u * h->geom_tris[geometry], num_vertices);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->master.dynpos[geometry], xyz, (size_t)num_vertices * 12, device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              h->stream));
    if (!device_src) HIP_TRY(h, hipStreamSynchronize(h->stream)); // the host array is only borrowed for the call
    h->master.mesh_dirty[h->geom_mesh[geometry]] = 1;
    h->vertex_updates++;
    return RPTR_OK;
}
int rptr_hip_update_vertices(rptr_hip_t *h, uint32_t geometry, const float *xyz, uint32_t num_vertices) {
    return update_vertices_common(h, geometry, xyz, num_vertices, false);
}
// the reference animates on the device (a compute shader writes float_vertex_buf, render_vulkan.cpp:2834-2840):
// same call with a DEVICE source, ordered on the backend's stream, no host synchronisation
int rptr_hip_update_vertices_device(rptr_hip_t *h, uint32_t geometry, const float *device_xyz, uint32_t num_vertices) {
    return update_vertices_common(h, geometry, device_xyz, num_vertices, true);
}

// ≙ BLAS update (VK_BUILD_ACCELERATION_STRUCTURE_MODE_UPDATE) of the dirty dynamic meshes + TLAS refit
// (render_vulkan.cpp:1323-1354, executed at the top of draw_frame :2165): topology is kept, triangles and all
// boxes are recomputed on the device, level by level from the leaves up.
extern "C++" {
// the shading records (dshade.h RpShadeTri) of one scene copy's triangles: of mesh `only_mesh`, or (-1) of every mesh
static int build_shade_records(rptr_hip *h, SceneCopy &sc, int only_mesh, hipStream_t st) {
    if (h->flat_tris && only_mesh < 0) // the world-space tree over the static instances' triangles: every triangle names its instance record
        hipLaunchKernelGGL(rp_k_build_shade_tris, dim3(grid_for(h, h->flat_tris)), dim3(256), 0, st, sc.dscene, sc.shade, 0u, (uint32_t)h->flat_tris, -1);
    for (size_t m = 0; m < h->meshes.size(); ++m) { // meshes with trees of their own (a mesh inside the flattened tree has none)
        const MeshRt &mr = h->meshes[m];
        if ((only_mesh >= 0 && (int)m != only_mesh) || mr.tri_count <= 0 || h->mesh_geometry_base[m] < 0) continue;
        hipLaunchKernelGGL(rp_k_build_shade_tris, dim3(grid_for(h, (size_t)mr.tri_count)), dim3(256), 0, st, sc.dscene, sc.shade, (uint32_t)mr.tri_base,
                           (uint32_t)mr.tri_count, h->mesh_geometry_base[m]);
    }
    HIP_TRY(h, hipGetLastError());
    return RPTR_OK;
}

// the depth levels of dynamic mesh m of one scene copy, deepest first: a launch per deep level, the shallow ones (at most 4^5 + ... + 1
// nodes) in the single-block kernel, which also does the instance bounds and the top level when `with_top`
static void refit_mesh_levels(rptr_hip *h, SceneCopy &sc, size_t m, hipStream_t st, bool with_top) {
    const MeshRt &mr = h->meshes[m];
    if (!sc.levels_known[m] && sc.ev_levels[m] && hipEventQuery(sc.ev_levels[m]) == hipSuccess) { // the read-back of a device-built tree's table has arrived
        memcpy(sc.host_levels[m].data(), sc.pinned_levels[m], RP_REFIT_LEVELS * sizeof(uint2));
        sc.levels_known[m] = 1;
    }
    const uint2 *dev_levels = sc.blas_levels + m * RP_REFIT_LEVELS;
    const int n_top = 6; // depths 0..5
    for (int slot = 0; slot < RP_REFIT_LEVELS - n_top; ++slot) {
        size_t work = (size_t)mr.node_capacity; // level size unknown to the host: any launch covers it (grid stride)
        if (sc.levels_known[m]) {
            work = sc.host_levels[m][(size_t)slot].y - sc.host_levels[m][(size_t)slot].x;
            if (!work) continue;
        }
        hipLaunchKernelGGL(rp_k_refit_level, dim3(grid_for(h, work, 4)), dim3(256), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.blas_list, dev_levels + slot);
    }
    RptrBvhInstance *insts = const_cast<RptrBvhInstance *>(sc.dscene.insts);
    hipLaunchKernelGGL(rp_k_refit_top, dim3(1), dim3(1024), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box, sc.blas_list,
                       dev_levels + (RP_REFIT_LEVELS - n_top), n_top, h->d_refit_list, h->d_refit_levels, with_top ? (int)h->refit_levels_tlas.size() : 0, insts,
                       with_top ? (uint32_t)h->num_tlas_insts : 0u);
}

// device-side rebuild of the bottom-level tree of dynamic mesh m of one scene copy (lbvh.h), on stream `st`. The triangles of the mesh
// (current order) must hold the new vertices already (rp_k_refit_tris). Ends with the refit that gives the new topology its boxes.
static int lbvh_rebuild(rptr_hip *h, SceneCopy &sc, size_t m, hipStream_t st, bool with_top) {
    const MeshRt &mr = h->meshes[m];
    const uint32_t n = (uint32_t)mr.tri_count;
    RpLbvhScratch &w = sc.scratch;
    if (w.capacity < (size_t)std::max<uint32_t>(n, 2)) { // first rebuild (of a mesh this large): work space for the largest dynamic mesh
        size_t cap = 2;
        for (const MeshRt &x : h->meshes)
            if (x.dynamic) cap = std::max<size_t>(cap, (size_t)x.tri_count);
        // the work space is allocated into a local record and committed as a whole: a failure half way frees what it got (the rebuild is
        // retried with every refit, and a retry must not leak the earlier attempt's buffers while the device is short of memory)
        RpLbvhScratch t = w;
        std::vector<void *> got;
        auto fail_alloc = [&](int code) {
            for (void *p : got) (void)hipFree(p);
            return code;
        };
        auto alloc = [&](auto **out, size_t count) -> int {
            void *p = nullptr;
            const size_t bytes = std::max<size_t>(count, 1) * sizeof(**out);
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return fail(h, RPTR_E_NOMEM, "hipMalloc(%zu) failed: %s (work space of a device-side BVH rebuild)", bytes, hipGetErrorString(e));
            got.push_back(p);
            *out = reinterpret_cast<std::remove_reference_t<decltype(**out)> *>(p);
            return RPTR_OK;
        };
        int rc;
        if ((rc = alloc(&t.keys_a, cap)) || (rc = alloc(&t.keys_b, cap))) return fail_alloc(rc);
        for (int **p : {&t.left, &t.right, &t.parent, &t.first, &t.last})
            if ((rc = alloc(p, cap))) return fail_alloc(rc);
        for (uint32_t **p : {&t.flag, &t.slot, &t.depth4})
            if ((rc = alloc(p, cap))) return fail_alloc(rc);
        if ((rc = alloc(&t.level_hist, RP_REFIT_LEVELS)) || (rc = alloc(&t.level_cursor, RP_REFIT_LEVELS)) || (rc = alloc(&t.tri_copy, cap)) ||
            (rc = alloc(&t.tribox_copy, 6 * cap)) || (rc = alloc(&t.bounds, 8)))
            return fail_alloc(rc);
        size_t sort_bytes = 0, scan_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, t.keys_a, t.keys_b, (int)cap, 0, 64, st);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, t.flag, t.slot, (int)cap, st);
        t.cub_bytes = std::max(sort_bytes, scan_bytes) + 256;
        char *tmp = nullptr;
        if ((rc = alloc(&tmp, t.cub_bytes))) return fail_alloc(rc);
        t.cub_tmp = tmp;
        t.capacity = cap;
        for (void *p : got) { // committed: the scene owns the buffers now (an earlier, smaller work space stays until the next set_scene)
            h->scene_allocs.push_back(p);
        }
        w = t;
    }
    RptrBvhTri *tris = sc.tris + mr.tri_base;
    float *tri_box = sc.tri_box + 6ull * mr.tri_base;
    const int g = grid_for(h, n);
    if (n >= 2) {
        hipLaunchKernelGGL(rp_k_lbvh_reset, dim3(1), dim3(64), 0, st, w.bounds);
        hipLaunchKernelGGL(rp_k_lbvh_bounds, dim3(g), dim3(256), 0, st, tri_box, n, w.bounds);
        int index_bits = 1;
        while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
        hipLaunchKernelGGL(rp_k_lbvh_keys, dim3(g), dim3(256), 0, st, tri_box, n, w.bounds, w.keys_a, index_bits);
        size_t bytes = w.cub_bytes;
        HIP_TRY(h, hipcub::DeviceRadixSort::SortKeys(w.cub_tmp, bytes, w.keys_a, w.keys_b, (int)n, 0, 64, st));
        hipLaunchKernelGGL(rp_k_lbvh_hierarchy, dim3(g), dim3(256), 0, st, w.keys_b, (int)n, w.left, w.right, w.parent, w.first, w.last);
        HIP_TRY(h, hipMemcpyAsync(w.tri_copy, tris, (size_t)n * sizeof(RptrBvhTri), hipMemcpyDeviceToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(w.tribox_copy, tri_box, (size_t)n * 24, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(rp_k_lbvh_gather, dim3(g), dim3(256), 0, st, w.keys_b, n, w.tri_copy, w.tribox_copy, tris, tri_box, (1ull << index_bits) - 1ull);
        hipLaunchKernelGGL(rp_k_lbvh_flags, dim3(g), dim3(256), 0, st, (int)n, w.parent, w.first, w.last, w.flag, w.depth4);
        bytes = w.cub_bytes;
        HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(w.cub_tmp, bytes, w.flag, w.slot, (int)n - 1, st));
    }
    HIP_TRY(h, hipMemsetAsync(w.level_hist, 0, RP_REFIT_LEVELS * sizeof(uint32_t), st));
    hipLaunchKernelGGL(rp_k_lbvh_emit, dim3(g), dim3(256), 0, st, (int)n, w.left, w.right, w.first, w.last, w.flag, w.slot, w.depth4, mr.node_base, mr.tri_base, sc.nodes,
                       w.level_hist, sc.mesh_count + m);
    uint2 *dev_levels = sc.blas_levels + m * RP_REFIT_LEVELS;
    hipLaunchKernelGGL(rp_k_lbvh_level_scan, dim3(1), dim3(64), 0, st, w.level_hist, (uint32_t)mr.node_base, dev_levels, w.level_cursor);
    hipLaunchKernelGGL(rp_k_lbvh_level_scatter, dim3(grid_for(h, (size_t)mr.node_capacity)), dim3(256), 0, st, sc.nodes, mr.node_base, sc.mesh_count + m, w.level_cursor,
                       sc.blas_list);
    // the host learns the level sizes when this copy has arrived; until then a refit launches every possible level
    sc.levels_known[m] = 0;
    HIP_TRY(h, hipMemcpyAsync(sc.pinned_levels[m], dev_levels, RP_REFIT_LEVELS * sizeof(uint2), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipEventRecord(sc.ev_levels[m], st));
    refit_mesh_levels(h, sc, m, st, with_top);
    HIP_TRY(h, hipGetLastError());
    sc.device_built[m] = 1;
    h->rebuilds_done++;
    return RPTR_OK;
}

// refits one copy of the mutable scene on stream `st`; all_dynamic: treat every dynamic mesh as changed. A mesh whose tree is older
// than the rebuild the policy asked for (rptr_hip_refit) is rebuilt instead of refitted. A rebuild that cannot start (no memory for its
// work space) is reported through *err -- the error text is in the handle -- and the mesh is refitted on its old topology instead, so
// that its boxes always match the new vertices; the rebuild is tried again with the next refit.
static bool refit_scene_copy(rptr_hip *h, SceneCopy &sc, bool all_dynamic, hipStream_t st, int *err) {
    bool any = all_dynamic && h->has_dynamic;
    for (size_t m = 0; m < h->meshes.size(); ++m) any = any || sc.mesh_dirty[m] == 1 || (h->meshes[m].dynamic && sc.built_epoch[m] != h->rebuild_epoch[m]);
    if (!any) return false;
    std::vector<size_t> todo;
    for (size_t m = 0; m < h->meshes.size(); ++m) {
        const MeshRt &mr = h->meshes[m];
        if (!mr.dynamic) continue;
        const bool rebuild = sc.built_epoch[m] != h->rebuild_epoch[m];
        if (!all_dynamic && !sc.mesh_dirty[m] && !rebuild) continue; // 1 = new vertices, 2 = dynamic but its triangle bounds were never written
        todo.push_back(m);
    }
    // the instance bounds and the top level ride in the single-block launch of the last mesh when they are small
    bool top_done = false;
    for (size_t k = 0; k < todo.size(); ++k) {
        const size_t m = todo[k];
        const MeshRt &mr = h->meshes[m];
        const bool with_top = h->refit_top_all && k + 1 == todo.size();
        if (mr.tri_count)
            hipLaunchKernelGGL(rp_k_refit_tris, dim3(grid_for(h, (size_t)mr.tri_count)), dim3(256), 0, st, sc.tris, sc.tri_box, sc.shade, (uint32_t)mr.tri_base,
                               (uint32_t)mr.tri_count, sc.mesh_dyn[m]);
        sc.mesh_dirty[m] = 0;
        if (sc.built_epoch[m] != h->rebuild_epoch[m]) {
            const int rc = lbvh_rebuild(h, sc, m, st, with_top);
            if (rc == RPTR_OK) {
                sc.built_epoch[m] = h->rebuild_epoch[m];
                (void)build_shade_records(h, sc, (int)m, st); // the rebuild reordered the mesh's triangles: its shading records follow
            } else {
                if (err && *err == RPTR_OK) *err = rc;
                refit_mesh_levels(h, sc, m, st, with_top);
            }
        } else
            refit_mesh_levels(h, sc, m, st, with_top);
        top_done = top_done || with_top;
    }
    if (!top_done) { // instance bounds, then the top level
        RptrBvhInstance *insts = const_cast<RptrBvhInstance *>(sc.dscene.insts);
        const uint32_t ni = (uint32_t)h->num_tlas_insts;
        if (h->refit_top_all)
            hipLaunchKernelGGL(rp_k_refit_top, dim3(1), dim3(1024), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box, sc.blas_list, sc.blas_levels, 0,
                               h->d_refit_list, h->d_refit_levels, (int)h->refit_levels_tlas.size(), insts, ni);
        else {
            if (ni) hipLaunchKernelGGL(rp_k_refit_instances, dim3(grid_for(h, ni)), dim3(256), 0, st, sc.node_box, insts, sc.inst_box, ni);
            for (auto &lv : h->refit_levels_tlas)
                hipLaunchKernelGGL(rp_k_refit_nodes, dim3(grid_for(h, lv[1] - lv[0])), dim3(256), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box,
                                   h->d_refit_list, lv[0], lv[1]);
        }
    }
    return true;
}
} // extern "C++"

int rptr_hip_refit(rptr_hip_t *h) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "refit before set_scene");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->ctx_scene.empty()) { // frames in flight read the master set
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    // ---- the BVH policy: which dynamic meshes get a new tree instead of a refit (librender/render_params.glsl.h:61,90-93)
    {
        bool changed = false;
        for (size_t m = 0; m < h->meshes.size(); ++m) changed = changed || (h->meshes[m].dynamic && h->master.mesh_dirty[m] == 1);
        if (changed && h->bvh_force_rebuild) {
            for (size_t m = 0; m < h->meshes.size(); ++m)
                if (h->meshes[m].rebuildable && h->master.mesh_dirty[m] == 1) h->rebuild_epoch[m]++;
        } else if (changed && h->bvh_budget > 0) {
            // a budget of triangles per refit call: it is saved up until it covers the next mesh in turn (a mesh larger than the budget is
            // rebuilt every ceil(triangles / budget) calls), dynamic meshes take turns
            long long total = 0;
            std::vector<size_t> dyn;
            for (size_t m = 0; m < h->meshes.size(); ++m)
                if (h->meshes[m].rebuildable) {
                    dyn.push_back(m);
                    total += h->meshes[m].tri_count;
                }
            h->bvh_credit = std::min(h->bvh_credit + h->bvh_budget, std::max(total, h->bvh_budget));
            for (size_t tries = 0; tries < dyn.size() && !dyn.empty(); ++tries) {
                const size_t m = dyn[(size_t)h->rebuild_cursor % dyn.size()];
                if (h->bvh_credit < h->meshes[m].tri_count) break;
                h->bvh_credit -= h->meshes[m].tri_count;
                h->rebuild_epoch[m]++;
                h->rebuild_cursor = (h->rebuild_cursor + 1) % (int)dyn.size();
            }
        }
    }
    if (!h->ctx_scene.empty()) {
        // frames render from the contexts' own sets, which follow from the master's VERTICES when their next frame is submitted:
        // the master's tree is only needed by ray queries and the export, and is refitted when one of them asks for it
        if (h->vertex_updates != h->vertex_updates_refitted) { // (the dirty marks stay for the deferred refit of the master tree)
            h->vertex_updates_refitted = h->vertex_updates;
            h->master_refit_pending = true;
            h->host_bvh_stale = true;
            h->refit_version++;
        }
        return RPTR_OK;
    }
    int err = RPTR_OK;
    if (refit_scene_copy(h, h->master, false, h->stream, &err)) {
        HIP_TRY(h, hipGetLastError());
        h->host_bvh_stale = true;
        h->refit_version++; // the frame contexts' own sets follow when their next frame is submitted
        h->master.version = h->refit_version;
    }
    return err;
}

// the master set's tree after a deferred refit (see rptr_hip_refit)
static int ensure_master_tree(rptr_hip *h) {
    if (!h->master_refit_pending) return RPTR_OK;
    h->master_refit_pending = false;
    int err = RPTR_OK;
    if (refit_scene_copy(h, h->master, false, h->stream, &err)) HIP_TRY(h, hipGetLastError());
    h->master.version = h->refit_version;
    return err;
}

