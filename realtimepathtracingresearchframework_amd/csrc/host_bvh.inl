// host_bvh.inl -- the acceleration structure's host side (rptr_hip_set_scene's first half; no device needed except for the device builder):
// scene validation, per-mesh binned-SAH trees / the device's PLOC builder, the top level, flattening, the 4-wide collapse and encoding
// Part of the ONE translation unit rptr_hip.hip (included there, in this order: host_state.h, host_bvh.inl, host_scene.inl,
// host_frame.inl, host_access.inl, host_comm.h): the host runtime split along its seams; no symbol changed.
extern "C++" {
// ------------------------------------------------------------------ host side of the acceleration structure
// Everything of set_scene that needs no device: per-mesh binned-SAH trees from the dequantised triangles, the
// top level over the instance bounds, the 4-wide collapse, the 64-byte encoding, and the worst-case stack need.
// Also reachable without a GPU through rptr_hip_build_bvh_host (CPU tests walk this tree with the oracle).
struct HostBvh {
    std::vector<RptrBvh4Node> nodes;
    std::vector<std::array<float, 6>> node_box;
    std::vector<RptrBvhTri> tris;
    std::vector<RptrBvhInstance> insts;
    std::vector<MeshRt> meshes;
    std::vector<int> mesh_root;
    int num_tlas_nodes = 0;
    int num_tlas_insts = 0; // instance records the top level refers to (a flattened scene keeps the scene's own records behind them)
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {1, 1, 1};
    int stack_need = 0;
    size_t flat_tris = 0, flat_nodes = 0; // a (partially) flattened scene: triangles / nodes of its one world-space tree (they come first)
    int flat_id_bias = 0;                 // ... and where its triangles' own instance records start (record = bias + instance id)
    bool device_built = false; // some bottom-level tree came from the device builder (ploc.h)
    double device_ms = 0.0;
    int device_iterations = 0;
};

// Flattening (option "flatten", default auto): a static scene with several instances is built as ONE bottom-level tree over all instanced triangles,
// pre-transformed to world space (a 10 M-triangle forest is 0.6 GB of triangles and nodes: nothing on a 288 GB device). Rays then
// meet one well-separated tree instead of a thousand overlapping instance boxes, each with its own ray transform. Hits are
// found on the world-space triangles, so t / u / v may differ from the two-level walk by rounding; shading still reads the
// mesh's own vertex streams through the instance record the triangle names (RptrBvhTri.flags bits 8..31).
// What both set_scene and rptr_hip_build_bvh_host check before they touch the borrowed arrays: index ranges of the mesh /
// geometry / material tables (a malformed .vks file must be rejected, not read out of bounds). Returns "" when fine.
static std::string validate_scene_tables(const RptrSceneDesc *s) {
    char buf[256];
    auto err = [&](const char *fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        return std::string(buf);
    };
    if ((s->num_geometries && !s->geometries) || (s->num_meshes && !s->meshes) || (s->num_parameterized_meshes && !s->parameterized_meshes) ||
        (s->num_instances && !s->instances) || (s->num_materials && !s->materials) || (s->num_lights && !s->lights))
        return "a table of the scene is NULL but its count is not 0";
    for (uint32_t g = 0; g < s->num_geometries; ++g)
        if (s->geometries[g].num_tris && !s->geometries[g].qpos) return err("geometry %u: qpos is NULL", g);
    for (uint32_t m = 0; m < s->num_meshes; ++m)
        if ((uint64_t)s->meshes[m].first_geometry + s->meshes[m].num_geometries > s->num_geometries)
            return err("mesh %u: geometries [%u, +%u) are outside the scene's %u geometries", m, s->meshes[m].first_geometry, s->meshes[m].num_geometries,
                       s->num_geometries);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        if (pm.mesh >= s->num_meshes) return err("parameterized mesh %u: bad mesh index", p);
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        if (mesh.num_geometries && !pm.material_offsets) return err("parameterized mesh %u: material_offsets is NULL", p);
        size_t off = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t nt = s->geometries[mesh.first_geometry + j].num_tris;
            if (pm.material_offsets[j] < 0 || (uint32_t)pm.material_offsets[j] >= s->num_materials)
                return err("parameterized mesh %u geometry %u: material offset %d out of range (%u materials)", p, j, pm.material_offsets[j], s->num_materials);
            if (pm.tri_material_ids) {
                uint32_t max_local = 0;
                for (uint32_t t = 0; t < nt; ++t) max_local = std::max<uint32_t>(max_local, pm.tri_material_ids[off + t]);
                if (nt && (uint64_t)pm.material_offsets[j] + max_local >= s->num_materials)
                    return err("parameterized mesh %u geometry %u: per-triangle material id %u + offset %d is outside the scene's %u materials", p, j, max_local,
                               pm.material_offsets[j], s->num_materials);
            }
            off += nt;
        }
    }
    for (uint32_t i = 0; i < s->num_instances; ++i)
        if (s->instances[i].parameterized_mesh >= s->num_parameterized_meshes) return err("instance %u: bad mesh", i);
    return std::string();
}

// 0: two-level; 1: the whole scene is one world-space tree (every instanced mesh is static); 2: PARTIAL -- the scene has dynamic meshes: the
// instances of its static meshes are flattened into one tree, which the top level holds as one identity instance beside the records of the
// dynamic meshes' instances (round 5: a forest with one animated character used to fall back to the two-level walk as a whole: 1.5 x)
static int want_flatten(const RptrSceneDesc *s, const RpOptions &o) {
    // option "flatten": -1 / 1 = every static multi-instance scene that fits "flatten_max_tris" (the default: the library knows which
    // meshes are dynamic -- RptrMeshDesc.dynamic, the reference's per-mesh build intent, vulkan/render_vulkan.cpp:942-952 -- and a flattened
    // tree is 1.5-1.6 x faster to trace than the two-level one); 0 = never
    if (o.v[OPT_FLATTEN] == 0 || s->num_instances < 2) return 0;
    const size_t limit = (size_t)o.v[OPT_FLATTEN_MAX_TRIS];
    size_t total = 0;
    uint32_t n_static = 0, n_dynamic = 0;
    for (uint32_t i = 0; i < s->num_instances; ++i) {
        const RptrMeshDesc &mesh = s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh].mesh];
        if (mesh.dynamic) {
            ++n_dynamic;
            continue;
        }
        ++n_static;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) total += s->geometries[mesh.first_geometry + j].num_tris;
    }
    if (total > limit || total == 0 || (uint64_t)s->num_instances + n_dynamic + 2 >= (1u << 24)) return 0;
    if (n_dynamic == 0) return 1;
    return n_static >= 2 ? 2 : 0;
}

// ------------------------------------------------------------------ device-side build of one bottom-level tree (ploc.h)
// What a build hands back to build_host_bvh: the tree in the form the host builder's encode_tree produces (local node indices from 0,
// leaf ranges from triangle 0 of `tris`), so that everything behind it -- top level, relocation, stack need, upload -- is shared.
struct DeviceTree {
    std::vector<RptrBvh4Node> nodes;
    std::vector<std::array<float, 6>> boxes;
    std::vector<RptrBvhTri> tris;
    double ms_device = 0.0, ms_top = 0.0;
    int iterations = 0;
    uint32_t top_clusters = 0;
};
// segments: the triangles' sources (device pointers of the vertex streams the scene upload made); mat_alpha: per material, 1 = alpha-tested
using DeviceTreeBuilder = std::function<bool(const std::vector<RpBuildSegment> &, uint32_t, DeviceTree &)>;
struct DeviceBuildCtx {
    DeviceTreeBuilder build;                         // empty: no device (rptr_hip_build_bvh_host on a CPU box)
    const std::vector<const uint64_t *> *d_qpos = nullptr; // per global geometry
    const std::vector<RpGeomRecord> *geoms = nullptr; // per (parameterized mesh, geometry): mat_ids
    size_t min_tris = (size_t)2 << 20;               // RPTR_BVH_BUILDER=auto: prim sets of at least this size are built on the device
};

namespace {
struct DevScratch { // frees what it allocated when the build is over
    std::vector<void *> ptrs;
    template <class T>
    T *get(size_t count) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T *)p;
    }
    ~DevScratch() {
        for (void *p : ptrs) (void)hipFree(p);
    }
};
} // namespace

static bool device_build_tree(rptr_hip *h, const std::vector<RpBuildSegment> &segs, uint32_t n, const std::vector<uint8_t> &mat_alpha, DeviceTree &out) {
    if (n < 2 || segs.empty()) return false;
    hipStream_t st = h->stream;
    DevScratch S;
#define DB_TRY(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            fail(h, RPTR_E_HIP, "device BVH build: %s failed: %s", #expr, hipGetErrorString(_e));    \
            (void)hipGetLastError();                                                                  \
            return false;                                                                             \
        }                                                                                             \
    } while (0)
#define DB_ALLOC(var, T, count)                                                          \
    T *var = S.get<T>(count);                                                            \
    if (!var) {                                                                          \
        fail(h, RPTR_E_NOMEM, "device BVH build: out of device memory (%s)", #var);     \
        (void)hipGetLastError();                                                         \
        return false;                                                                    \
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct EvGuard {
        hipEvent_t &a, &b;
        ~EvGuard() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev_guard{e0, e1};
    (void)hipEventRecord(e0, st);
    const int g = grid_for(h, n);
    // 1. triangles + bounds
    DB_ALLOC(d_segs, RpBuildSegment, segs.size());
    DB_ALLOC(d_alpha, uint8_t, mat_alpha.size());
    DB_ALLOC(tris_a, RptrBvhTri, (size_t)n + 2);
    DB_ALLOC(tris_b, RptrBvhTri, (size_t)n + 2);
    DB_ALLOC(box_a, float, 6 * (size_t)n);
    DB_ALLOC(box_b, float, 6 * (size_t)n);
    DB_TRY(hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(RpBuildSegment), hipMemcpyHostToDevice, st));
    if (!mat_alpha.empty()) DB_TRY(hipMemcpyAsync(d_alpha, mat_alpha.data(), mat_alpha.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rp_k_build_tris, dim3(g), dim3(256), 0, st, d_segs, (int)segs.size(), n, d_alpha, (uint32_t)mat_alpha.size(), tris_a, box_a);
    // 2. Morton order
    DB_ALLOC(keys_a, unsigned long long, n);
    DB_ALLOC(keys_b, unsigned long long, n);
    DB_ALLOC(bounds, uint32_t, 8);
    size_t sort_bytes = 0, scan_bytes = 0, scan64_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_a, keys_b, (int)n, 0, 64, st);
    DB_ALLOC(flag, uint32_t, n);
    DB_ALLOC(slot, uint32_t, n);
    DB_ALLOC(packed, unsigned long long, n);
    DB_ALLOC(pscan, unsigned long long, n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flag, slot, (int)n, st);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan64_bytes, packed, pscan, (int)n, st);
    const size_t cub_bytes = std::max(sort_bytes, std::max(scan_bytes, scan64_bytes)) + 256;
    DB_ALLOC(cub_tmp, char, cub_bytes);
    hipLaunchKernelGGL(rp_k_lbvh_reset, dim3(1), dim3(64), 0, st, bounds);
    hipLaunchKernelGGL(rp_k_lbvh_bounds, dim3(g), dim3(256), 0, st, box_a, n, bounds);
    int index_bits = 1;
    while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
    hipLaunchKernelGGL(rp_k_lbvh_keys, dim3(g), dim3(256), 0, st, box_a, n, bounds, keys_a, index_bits);
    size_t bytes = cub_bytes;
    DB_TRY(hipcub::DeviceRadixSort::SortKeys(cub_tmp, bytes, keys_a, keys_b, (int)n, 0, 64, st));
    hipLaunchKernelGGL(rp_k_lbvh_gather, dim3(g), dim3(256), 0, st, keys_b, n, tris_a, box_a, tris_b, box_b, (1ull << index_bits) - 1ull);
    // 3. PLOC
    DB_ALLOC(cid_a, uint32_t, n);
    DB_ALLOC(cid_b, uint32_t, n);
    DB_ALLOC(nn, uint32_t, n);
    DB_ALLOC(left, int, n);
    DB_ALLOC(right, int, n);
    DB_ALLOC(parent, int, 2 * (size_t)n);
    DB_ALLOC(count, uint32_t, 2 * (size_t)n);
    DB_ALLOC(area, float, n);
    DB_ALLOC(totals, uint32_t, 4);
    DB_ALLOC(dpc, float4, n); // per inner node: the costs the collapse decides by (ploc.h rp_ploc_dp_node)
    float *cbox_a = box_a, *cbox_b = nullptr; // (box_a is free again after the gather; the second cluster box list is its own)
    DB_ALLOC(cbox_second, float, 6 * (size_t)n);
    cbox_b = cbox_second;
    hipLaunchKernelGGL(rp_k_ploc_init, dim3(g), dim3(256), 0, st, n, box_b, cid_a, cbox_a, parent, count);
    uint32_t host_totals[2] = {n, n}; // clusters, nodes made so far (ids below n are the triangles)
    DB_TRY(hipMemcpyAsync(totals, host_totals, sizeof(host_totals), hipMemcpyHostToDevice, st));
    size_t top_k = RP_PLOC_TOP;
    if (h->opt.v[OPT_PLOC_TOP] > 0) top_k = (size_t)h->opt.v[OPT_PLOC_TOP];
    uint32_t m = n, nodes_before = n;
    int iterations = 0;
    while (m > top_k && m > 1) {
        hipLaunchKernelGGL(rp_k_ploc_nn<RP_PLOC_RADIUS>, dim3((m + 255) / 256), dim3(256), 0, st, m, cbox_a, nn);
        hipLaunchKernelGGL(rp_k_ploc_flags, dim3(grid_for(h, m)), dim3(256), 0, st, m, nn, packed);
        bytes = cub_bytes;
        DB_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes, packed, pscan, (int)m, st));
        hipLaunchKernelGGL(rp_k_ploc_apply, dim3(grid_for(h, m)), dim3(256), 0, st, m, n, nn, packed, pscan, cid_a, cbox_a, cid_b, cbox_b, left, right, parent, count, area, totals,
                           totals + 2);
        DB_TRY(hipMemcpyAsync(totals, totals + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        DB_TRY(hipMemcpyAsync(host_totals, totals, sizeof(host_totals), hipMemcpyDeviceToHost, st));
        DB_TRY(hipStreamSynchronize(st));
        if (host_totals[0] >= m) { // (cannot happen: the globally closest pair is always mutual)
            fail(h, RPTR_E_HIP, "device BVH build: clustering made no progress at %u clusters", m);
            return false;
        }
        if (host_totals[1] > nodes_before) // the merges of this iteration: their children are older, their costs known
            hipLaunchKernelGGL(rp_k_ploc_dp_range, dim3(grid_for(h, host_totals[1] - nodes_before)), dim3(256), 0, st, nodes_before, host_totals[1], n, left, right, count, area, dpc);
        nodes_before = host_totals[1];
        m = host_totals[0];
        std::swap(cid_a, cid_b);
        std::swap(cbox_a, cbox_b);
        ++iterations;
    }
    out.iterations = iterations;
    out.top_clusters = m;
    // 4. the top: binned SAH over the remaining clusters (host, milliseconds), stitched on
    const auto t_top0 = std::chrono::steady_clock::now();
    if (m > 1) {
        std::vector<rptr::BuildPrim> cp(m);
        std::vector<uint32_t> ids(m), cnt(m);
        static_assert(sizeof(rptr::BuildPrim) == 24, "cluster boxes are copied as build primitives");
        DB_ALLOC(d_cnt, uint32_t, m);
        hipLaunchKernelGGL(rp_k_ploc_gather_counts, dim3(grid_for(h, m)), dim3(256), 0, st, m, cid_a, count, d_cnt);
        DB_TRY(hipMemcpyAsync(cp.data(), cbox_a, (size_t)m * 24, hipMemcpyDeviceToHost, st));
        DB_TRY(hipMemcpyAsync(ids.data(), cid_a, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        DB_TRY(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        DB_TRY(hipStreamSynchronize(st));
        rptr::BuiltTree top;
        rptr::build_bvh2(cp.data(), m, 1, 56, 0, top);
        const size_t T = top.nodes.size();
        if (T != (size_t)m - 1) {
            fail(h, RPTR_E_HIP, "device BVH build: the top tree over %u clusters has %zu nodes", m, T);
            return false;
        }
        std::vector<float4> ccost(m), tcost(T); // collapse costs of the cluster roots (from the device) and of the top nodes (made here)
        {
            DB_ALLOC(d_ccost, float4, m);
            hipLaunchKernelGGL(rp_k_ploc_gather_costs, dim3(grid_for(h, m)), dim3(256), 0, st, m, n, cid_a, dpc, d_ccost);
            DB_TRY(hipMemcpyAsync(ccost.data(), d_ccost, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, st));
            DB_TRY(hipStreamSynchronize(st));
        }
        std::vector<int> tl(T), tr(T);
        std::vector<uint32_t> tc(T);
        std::vector<float> ta(T);
        std::vector<float4> cost_of(T); // by top-tree node index
        std::vector<uint32_t> id_of(T), cnt_of(T);
        const uint32_t first_id = host_totals[1];
        for (int64_t i = (int64_t)T - 1; i >= 0; --i) { // children lie behind their parents: backwards = bottom-up, the root is made last
            const RptrBvhNode &t = top.nodes[(size_t)i];
            const size_t k = T - 1 - (size_t)i;
            uint32_t c_id[2], c_cnt[2];
            const int32_t two[2] = {t.child0, t.child1};
            for (int w = 0; w < 2; ++w) {
                if (two[w] >= 0) {
                    c_id[w] = id_of[(size_t)two[w]];
                    c_cnt[w] = cnt_of[(size_t)two[w]];
                } else {
                    const uint32_t ci = top.order[(size_t)RPTR_BVH_LEAF_FIRST(two[w])];
                    c_id[w] = ids[ci];
                    c_cnt[w] = cnt[ci];
                }
            }
            tl[k] = (int)c_id[0];
            tr[k] = (int)c_id[1];
            tc[k] = c_cnt[0] + c_cnt[1];
            {   // surface (half) area of the node's box = union of its children's boxes, as the clustering computes it for its own nodes
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::fmin(t.lo0[a], t.lo1[a]);
                    hi[a] = std::fmax(t.hi0[a], t.hi1[a]);
                }
                const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                ta[k] = (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? dx * dy + dy * dz + dz * dx : 0.0f;
            }
            {   // ploc.h rp_ploc_dp_node, for a node of the top (same operations in the same order)
                float4 cc[2];
                bool exp[2];
                for (int w = 0; w < 2; ++w) {
                    cc[w] = two[w] >= 0 ? cost_of[(size_t)two[w]] : ccost[top.order[(size_t)RPTR_BVH_LEAF_FIRST(two[w])]];
                    exp[w] = c_id[w] >= n && c_cnt[w] > (uint32_t)RP_LBVH_LEAF_TRIS;
                }
                auto G = [&](int w, int q) { return !exp[w] ? 0.0f : q == 1 ? cc[w].x : std::fmin(cc[w].x, q == 2 ? cc[w].y : q == 3 ? cc[w].z : cc[w].w); };
                const float f2 = G(0, 1) + G(1, 1), f3 = std::fmin(G(0, 1) + G(1, 2), G(0, 2) + G(1, 1)),
                            f4 = std::fmin(std::fmin(G(0, 1) + G(1, 3), G(0, 2) + G(1, 2)), G(0, 3) + G(1, 1));
                tcost[k] = cost_of[(size_t)i] = make_float4(ta[k] + f4, f2, f3, f4);
            }
            id_of[(size_t)i] = first_id + (uint32_t)k;
            cnt_of[(size_t)i] = tc[k];
        }
        DB_ALLOC(d_tl, int, T);
        DB_ALLOC(d_tr, int, T);
        DB_ALLOC(d_tc, uint32_t, T);
        DB_ALLOC(d_ta, float, T);
        DB_TRY(hipMemcpyAsync(d_ta, ta.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tl, tl.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tr, tr.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tc, tc.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(dpc + (first_id - n), tcost.data(), T * sizeof(float4), hipMemcpyHostToDevice, st)); // (top node k has id first_id + k)
        hipLaunchKernelGGL(rp_k_ploc_stitch, dim3(grid_for(h, T)), dim3(256), 0, st, (uint32_t)T, n, first_id, d_tl, d_tr, d_tc, d_ta, left, right, parent, count, area);
        DB_TRY(hipStreamSynchronize(st)); // (the host arrays are read by the copies above)
        if (first_id + (uint32_t)T != 2u * n - 1u) {
            fail(h, RPTR_E_HIP, "device BVH build: %u + %zu nodes for %u triangles", first_id, T, n);
            return false;
        }
    } else if (host_totals[1] != 2u * n - 1u) {
        fail(h, RPTR_E_HIP, "device BVH build: %u nodes for %u triangles", host_totals[1], n);
        return false;
    }
    out.ms_top = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_top0).count();
    // 5. depth-first order of the triangles (every subtree a contiguous range)
    const uint32_t root_id = 2u * n - 2u;
    DB_ALLOC(nfirst, uint32_t, n);
    hipLaunchKernelGGL(rp_k_ploc_firsts, dim3(g), dim3(256), 0, st, n, left, right, parent, count, nfirst);
    float *tri_box = cbox_b; // (the cluster box lists are dead after the stitch: one of them takes the triangle bounds in their final order)
    hipLaunchKernelGGL(rp_k_ploc_scatter, dim3(g), dim3(256), 0, st, n, left, right, parent, count, tris_b, box_b, tris_a, tri_box);
    // 6. 4-wide collapse, breadth first: a launch pair + one scan per depth level; then boxes + encoding, deepest level first
    DB_ALLOC(nodes, RptrBvh4Node, n);
    DB_ALLOC(node_box, float, 6 * (size_t)n);
    DB_ALLOC(queue_a, int, n);
    DB_ALLOC(queue_b, int, n);
    DB_ALLOC(d_next, uint32_t, 1);
    std::vector<uint32_t> level_base;
    {
        const int h_root = (int)root_id;
        DB_TRY(hipMemcpyAsync(queue_a, &h_root, sizeof(int), hipMemcpyHostToDevice, st));
        uint32_t size = 1, base = 0;
        while (size > 0) {
            if (level_base.size() >= 2 * RP_REFIT_LEVELS || (size_t)base + size > (size_t)n) {
                fail(h, RPTR_E_UNSUPPORTED, "device BVH build: a tree of more than %d levels / %u nodes", 2 * RP_REFIT_LEVELS, base + size);
                return false;
            }
            level_base.push_back(base);
            const int gl = grid_for(h, size);
            hipLaunchKernelGGL(rp_k_ploc_collapse_count, dim3(gl), dim3(256), 0, st, queue_a, size, n, left, right, count, dpc, flag);
            bytes = cub_bytes;
            DB_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes, flag, slot, (int)size, st));
            hipLaunchKernelGGL(rp_k_ploc_collapse_emit, dim3(gl), dim3(256), 0, st, queue_a, size, n, left, right, parent, count, dpc, nfirst, slot, base, base + size, nodes,
                               queue_b, d_next);
            uint32_t next = 0;
            DB_TRY(hipMemcpyAsync(&next, d_next, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DB_TRY(hipStreamSynchronize(st));
            base += size;
            size = next;
            std::swap(queue_a, queue_b);
        }
        level_base.push_back(base); // = the number of nodes
    }
    const int h_count = (int)level_base.back();
    for (size_t l = level_base.size() - 1; l-- > 0;)
        hipLaunchKernelGGL(rp_k_ploc_refit_range, dim3(grid_for(h, level_base[l + 1] - level_base[l])), dim3(256), 0, st, nodes, node_box, tri_box, level_base[l], level_base[l + 1]);
    (void)hipEventRecord(e1, st);
    // back to the host, in the host builder's form
    out.nodes.resize((size_t)h_count);
    out.boxes.resize((size_t)h_count);
    out.tris.resize(n);
    DB_TRY(hipMemcpyAsync(out.nodes.data(), nodes, (size_t)h_count * sizeof(RptrBvh4Node), hipMemcpyDeviceToHost, st));
    DB_TRY(hipMemcpyAsync(out.boxes.data(), node_box, (size_t)h_count * 24, hipMemcpyDeviceToHost, st));
    DB_TRY(hipMemcpyAsync(out.tris.data(), tris_a, (size_t)n * sizeof(RptrBvhTri), hipMemcpyDeviceToHost, st));
    DB_TRY(hipStreamSynchronize(st));
    DB_TRY(hipGetLastError());
    float ms = 0.f;
    if (e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) out.ms_device = ms;
    for (RptrBvh4Node &nd : out.nodes) nd._pad1[0] = 0; // (the depth parked there by the emit kernel is not part of the tree)
    return true;
#undef DB_TRY
#undef DB_ALLOC
}

static void build_host_bvh(const RptrSceneDesc *s, HostBvh &B, const RpOptions &opt, const DeviceBuildCtx *dev = nullptr) {
    // instanceCustomIndex of every parameterized mesh = number of geometries before it (render_vulkan.cpp:2748-2850)
    std::vector<int> pmesh_base(s->num_parameterized_meshes, 0);
    {
        int at = 0;
        for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
            pmesh_base[p] = at;
            at += (int)s->meshes[s->parameterized_meshes[p].mesh].num_geometries;
        }
    }
    // ---- bottom-level BVHs (one per mesh), built from dequantised floats
    B.nodes.clear();
    B.tris.clear();
    B.insts.clear();
    B.meshes.assign(s->num_meshes, MeshRt());
    std::vector<RptrBvh4Node> blas_nodes;          // relocated behind the TLAS afterwards
    std::vector<std::array<float, 6>> blas_boxes;  // exact float bounds per node (refit + instance bounds)
    // encodes a wide tree into 64-byte nodes; inner child indices get `node_shift`, leaf ranges `first_shift`
    auto encode_tree = [](const rptr::Wide4Tree &wt, int node_shift, int first_shift, std::vector<RptrBvh4Node> &dst,
                          std::vector<std::array<float, 6>> &boxes) {
        for (const rptr::Wide4 &w : wt.nodes) {
            int32_t child[4];
            for (int k = 0; k < 4; ++k) {
                const int32_t c = w.child[k];
                if (c == RPTR_BVH4_EMPTY)
                    child[k] = c;
                else if (c >= 0)
                    child[k] = c + node_shift;
                else
                    child[k] = RPTR_BVH_LEAF(RPTR_BVH_LEAF_FIRST(c) + first_shift, RPTR_BVH_LEAF_COUNT(c));
            }
            RptrBvh4Node n;
            std::array<float, 6> nb;
            rp_bvh4_encode(w.box, child, &n, nb.data(), nb.data() + 3);
            dst.push_back(n);
            boxes.push_back(nb);
        }
    };
    // candidates of the alpha test: a triangle is flagged when some parameterized mesh of its mesh assigns it a material
    // without BASE_MATERIAL_NOALPHA (the material is per parameterized mesh, the BLAS per mesh; the test itself looks
    // the material up again, kernels.h rp_alpha_rejects)
    std::vector<std::vector<uint8_t>> tri_alpha(s->num_meshes);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        std::vector<uint8_t> &fl = tri_alpha[pm.mesh];
        size_t off = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t nt = s->geometries[mesh.first_geometry + j].num_tris;
            if (fl.size() < off + nt) fl.resize(off + nt, 0);
            for (uint32_t t = 0; t < nt; ++t) {
                const int64_t mid = (int64_t)pm.material_offsets[j] + (pm.tri_material_ids ? (int64_t)pm.tri_material_ids[off + t] : 0);
                if (mid >= 0 && mid < (int64_t)s->num_materials && (s->materials[mid].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0) fl[off + t] = 1;
            }
            off += nt;
        }
    }
    const int flatten_mode = want_flatten(s, opt);
    const bool flatten = flatten_mode != 0, partial = flatten_mode == 2;
    // where a flattened triangle's OWN instance record lies in the instance array: behind the records the top level refers to (one for the
    // flat tree; PARTIAL: + one per instance of a dynamic mesh), at flat_bias + its instance id
    uint32_t n_dynamic_insts = 0;
    auto instance_is_dynamic = [&](uint32_t i) { return s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh].mesh].dynamic != 0; };
    for (uint32_t i = 0; i < s->num_instances; ++i) n_dynamic_insts += (partial && instance_is_dynamic(i)) ? 1u : 0u;
    const uint32_t flat_bias = 1u + n_dynamic_insts;
    B.flat_id_bias = flatten ? (int)flat_bias : 0;
    rptr::build_tuning().collapse_rule = (int)opt.v[OPT_COLLAPSE];
    rptr::build_tuning().ploc_top = (size_t)opt.v[OPT_PLOC_TOP];
    rptr::build_tuning().ploc_leaf = (int)opt.v[OPT_PLOC_LEAF];
    // spatial splits for static geometry (bvh_build.h presplit_triangles): RPTR_PRESPLIT="density[,budget]". Off unless asked for:
    // on the 10 M-triangle forest they buy 16 % fewer triangle tests for 7 % more node visits and twice the references
    // (profiles/r03_notes.md), on height fields nothing
    const float split_density = (float)opt.v[OPT_PRESPLIT_DENSITY], split_budget = (float)opt.v[OPT_PRESPLIT_BUDGET_PCT] * 0.01f;
    // who builds a bottom-level tree: option "bvh_builder" = 0 auto (the device for large static triangle sets, the host otherwise), 1 host, 2 device
    const int builder_mode = (int)opt.v[OPT_BVH_BUILDER];
    const int host_ploc = (int)opt.v[OPT_HOST_PLOC];
    auto on_device = [&](size_t n_tris) {
        return dev && dev->build && builder_mode != 1 && n_tris >= 2 && n_tris < ((size_t)1 << 28) && (builder_mode == 2 || n_tris >= dev->min_tris) &&
               !(split_density > 0.0f && split_budget > 0.0f);
    };
    bool flat_done = false;
    if (flatten) { // the one world-space tree on the device: triangles from the vertex streams, sort, clustering, collapse, encoding (ploc.h)
        size_t total = 0;
        std::vector<RpBuildSegment> segs;
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
            const RptrMeshDesc &mesh = s->meshes[pm.mesh];
            if (mesh.dynamic) continue; // (PARTIAL: its instances keep their own records and trees)
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                if (!gd.num_tris) continue;
                RpBuildSegment sg;
                memset(&sg, 0, sizeof(sg));
                if (dev && dev->d_qpos) {
                    sg.qpos = (*dev->d_qpos)[mesh.first_geometry + j];
                    sg.mat_ids = (*dev->geoms)[(size_t)pmesh_base[in.parameterized_mesh] + j].mat_ids;
                }
                memcpy(sg.scaling, gd.quantized_scaling, 12);
                memcpy(sg.offset, gd.quantized_offset, 12);
                sg.material_offset = pm.material_offsets[j];
                sg.begin = (uint32_t)total;
                memcpy(sg.transform, in.transform, 48);
                sg.count = gd.num_tris;
                sg.geom = j;
                sg.flags_hi = (flat_bias + i) << 8;
                sg.has_transform = 1;
                segs.push_back(sg);
                total += gd.num_tris;
            }
        }
        DeviceTree dt;
        if (on_device(total) && dev->build(segs, (uint32_t)total, dt)) {
            B.flat_tris = dt.tris.size();
            B.flat_nodes = dt.nodes.size();
            for (MeshRt &mr : B.meshes) {
                mr.node_base = 0;
                mr.node_count = 0;
                mr.tri_base = 0;
                mr.tri_count = 0;
                memcpy(mr.lo, dt.boxes[0].data(), 12);
                memcpy(mr.hi, dt.boxes[0].data() + 3, 12);
            }
            B.tris = std::move(dt.tris);
            blas_nodes = std::move(dt.nodes);
            blas_boxes = std::move(dt.boxes);
            B.device_built = true;
            B.device_ms = dt.ms_device;
            B.device_iterations = dt.iterations;
            flat_done = true;
        }
    }
    if (flatten && !flat_done) {
        std::vector<rptr::BuildPrim> prims;
        std::vector<RptrBvhTri> mtris;
        std::vector<rptr::TriVerts> verts;
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
            const RptrMeshDesc &mesh = s->meshes[pm.mesh];
            if (mesh.dynamic) continue;
            const float *M = in.transform;
            size_t off = 0;
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                for (uint32_t t = 0; t < gd.num_tris; ++t) {
                    float v[3][3], w[3][3];
                    for (int k = 0; k < 3; ++k) {
                        dequantize_position(gd.qpos[3 * (size_t)t + k], gd.quantized_scaling, gd.quantized_offset, v[k]);
                        for (int r = 0; r < 3; ++r) w[k][r] = ((M[4 * r] * v[k][0] + M[4 * r + 1] * v[k][1]) + M[4 * r + 2] * v[k][2]) + M[4 * r + 3];
                    }
                    RptrBvhTri tri;
                    rptr::BuildPrim bp;
                    for (int k = 0; k < 3; ++k) {
                        tri.v0[k] = w[0][k];
                        tri.e1[k] = w[1][k] - w[0][k];
                        tri.e2[k] = w[2][k] - w[0][k];
                        bp.lo[k] = std::fmin(w[0][k], std::fmin(w[1][k], w[2][k]));
                        bp.hi[k] = std::fmax(w[0][k], std::fmax(w[1][k], w[2][k]));
                    }
                    const int64_t mid = (int64_t)pm.material_offsets[j] + (pm.tri_material_ids ? (int64_t)pm.tri_material_ids[off + t] : 0);
                    const bool alpha = mid >= 0 && mid < (int64_t)s->num_materials && (s->materials[mid].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0;
                    tri.prim = t;
                    tri.geom = j;
                    tri.flags = (alpha ? RPTR_BVH_TRI_ALPHA : 0u) | ((flat_bias + i) << 8); // its instance: record flat_bias + i of the instance array
                    mtris.push_back(tri);
                    prims.push_back(bp);
                    rptr::TriVerts tv;
                    memcpy(tv.v, w, sizeof(tv.v));
                    verts.push_back(tv);
                }
                off += gd.num_tris;
            }
        }
        std::vector<uint32_t> ref_tri; // reference -> triangle (empty: one reference per triangle)
        if (split_density > 0.0f && split_budget > 0.0f) rptr::presplit_triangles(verts.data(), (uint32_t)verts.size(), split_density, split_budget, 256, 0, prims, ref_tri);
        std::vector<rptr::TriVerts>().swap(verts);
        rptr::BuiltTree tree;
        if (host_ploc > 0) // experiment: the clustering of the device builder, stated on the host
            rptr::build_bvh2_ploc(prims.data(), (uint32_t)prims.size(), host_ploc, RPTR_BVH_MAX_LEAF_TRIS, 0, tree);
        else
            rptr::build_bvh2(prims.data(), (uint32_t)prims.size(), RPTR_BVH_MAX_LEAF_TRIS, 48, 0, tree);
        rptr::Wide4Tree wide;
        rptr::collapse_bvh4(tree, wide);
        for (MeshRt &mr : B.meshes) { // no mesh has a tree of its own: they all point at the one tree
            mr.node_base = 0;
            mr.node_count = 0;
            mr.tri_base = 0;
            mr.tri_count = 0;
            memcpy(mr.lo, tree.lo, 12);
            memcpy(mr.hi, tree.hi, 12);
        }
        B.tris.reserve(tree.order.size());
        for (uint32_t id : tree.order) B.tris.push_back(mtris[ref_tri.empty() ? id : ref_tri[id]]);
        encode_tree(wide, 0, 0, blas_nodes, blas_boxes);
        B.flat_tris = B.tris.size();
        B.flat_nodes = blas_nodes.size();
    }
    for (uint32_t m = 0; m < s->num_meshes && (!flatten || partial); ++m) {
        const RptrMeshDesc &mesh = s->meshes[m];
        if (partial && !mesh.dynamic) continue; // (its instances are part of the flat tree)
        {
            size_t total = 0;
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) total += s->geometries[mesh.first_geometry + j].num_tris;
            if (mesh.dynamic == 0 && on_device(total) && dev->d_qpos) {
                std::vector<RpBuildSegment> segs;
                std::vector<size_t> geom_first(mesh.num_geometries, 0);
                size_t at = 0;
                for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                    const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                    geom_first[j] = at;
                    if (gd.num_tris) {
                        RpBuildSegment sg;
                        memset(&sg, 0, sizeof(sg));
                        sg.qpos = (*dev->d_qpos)[mesh.first_geometry + j];
                        memcpy(sg.scaling, gd.quantized_scaling, 12);
                        memcpy(sg.offset, gd.quantized_offset, 12);
                        sg.material_offset = -1; // (the alpha flag of a mesh's triangle is the OR over its parameterized meshes: set below)
                        sg.begin = (uint32_t)at;
                        sg.count = gd.num_tris;
                        sg.geom = j;
                        segs.push_back(sg);
                    }
                    at += gd.num_tris;
                }
                DeviceTree dt;
                if (dev->build(segs, (uint32_t)total, dt)) {
                    MeshRt &mr = B.meshes[m];
                    mr.dynamic = false;
                    mr.rebuildable = false;
                    mr.node_base = (int)blas_nodes.size();
                    mr.node_count = mr.node_capacity = (int)dt.nodes.size();
                    mr.tri_base = (int)B.tris.size();
                    mr.tri_count = (int)dt.tris.size();
                    memcpy(mr.lo, dt.boxes[0].data(), 12);
                    memcpy(mr.hi, dt.boxes[0].data() + 3, 12);
                    for (RptrBvhTri &t : dt.tris) {
                        const size_t lin = geom_first[t.geom] + t.prim;
                        if (lin < tri_alpha[m].size() && tri_alpha[m][lin]) t.flags |= RPTR_BVH_TRI_ALPHA;
                    }
                    for (RptrBvh4Node &nd : dt.nodes) // local -> absolute references (what encode_tree's shifts do for a host tree)
                        for (int k = 0; k < 4; ++k) {
                            const int32_t c = nd.child[k];
                            if (c == RPTR_BVH4_EMPTY) continue;
                            nd.child[k] = c >= 0 ? c + mr.node_base : RPTR_BVH_LEAF(RPTR_BVH_LEAF_FIRST(c) + mr.tri_base, RPTR_BVH_LEAF_COUNT(c));
                        }
                    B.tris.insert(B.tris.end(), dt.tris.begin(), dt.tris.end());
                    blas_nodes.insert(blas_nodes.end(), dt.nodes.begin(), dt.nodes.end());
                    blas_boxes.insert(blas_boxes.end(), dt.boxes.begin(), dt.boxes.end());
                    B.device_built = true;
                    B.device_ms += dt.ms_device;
                    B.device_iterations = std::max(B.device_iterations, dt.iterations);
                    continue;
                }
            }
        }
        std::vector<rptr::BuildPrim> prims;
        std::vector<RptrBvhTri> mtris;
        std::vector<rptr::TriVerts> verts;
        const bool split_mesh = mesh.dynamic == 0 && split_density > 0.0f && split_budget > 0.0f; // (a refit recomputes boxes from whole triangles)
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
            for (uint32_t t = 0; t < gd.num_tris; ++t) {
                float v[3][3];
                for (int k = 0; k < 3; ++k) dequantize_position(gd.qpos[3 * (size_t)t + k], gd.quantized_scaling, gd.quantized_offset, v[k]);
                RptrBvhTri tri;
                rptr::BuildPrim bp;
                for (int k = 0; k < 3; ++k) {
                    tri.v0[k] = v[0][k];
                    tri.e1[k] = v[1][k] - v[0][k];
                    tri.e2[k] = v[2][k] - v[0][k];
                    bp.lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k]));
                    bp.hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
                }
                tri.prim = t;
                tri.geom = j;
                tri.flags = (mtris.size() < tri_alpha[m].size() && tri_alpha[m][mtris.size()]) ? RPTR_BVH_TRI_ALPHA : 0u;
                mtris.push_back(tri);
                prims.push_back(bp);
                if (split_mesh) {
                    rptr::TriVerts tv;
                    memcpy(tv.v, v, sizeof(tv.v));
                    verts.push_back(tv);
                }
            }
        }
        MeshRt &mr = B.meshes[m];
        mr.dynamic = mesh.dynamic != 0;
        mr.rebuildable = mr.dynamic && (mesh.dynamic & RPTR_MESH_SUBTLY_DYNAMIC) == 0;
        std::vector<uint32_t> ref_tri;
        if (split_mesh) rptr::presplit_triangles(verts.data(), (uint32_t)verts.size(), split_density, split_budget, 256, 0, prims, ref_tri);
        rptr::BuiltTree tree;
        if (host_ploc > 0) // experiment: the clustering of the device builder, stated on the host
            rptr::build_bvh2_ploc(prims.data(), (uint32_t)prims.size(), host_ploc, RPTR_BVH_MAX_LEAF_TRIS, 0, tree);
        else
            rptr::build_bvh2(prims.data(), (uint32_t)prims.size(), RPTR_BVH_MAX_LEAF_TRIS, 48, 0, tree);
        rptr::Wide4Tree wide;
        // (the meshes of a scene whose instances get several sub-roots each -- partial re-braiding below -- keep the greedy collapse: the cut
        // through the top of the tree wants the balanced nodes it makes; with the area-optimal collapse the instanced forest needs 38.5
        // instead of 36.8 node visits per ray)
        rptr::collapse_bvh4(tree, wide, -1, (s->num_instances >= 16 && host_ploc <= 0) ? rptr::COLLAPSE_GREEDY : rptr::COLLAPSE_OPTIMAL); // (the device builder and its host statement: always the optimal one)
        mr.node_base = (int)blas_nodes.size();
        mr.node_count = (int)wide.nodes.size();
        mr.node_capacity = mr.dynamic ? std::max(mr.node_count, (int)mtris.size()) : mr.node_count;
        mr.tri_base = (int)B.tris.size();
        mr.tri_count = (int)tree.order.size(); // references (= triangles unless the mesh was pre-split)
        memcpy(mr.lo, tree.lo, 12);
        memcpy(mr.hi, tree.hi, 12);
        for (uint32_t id : tree.order) B.tris.push_back(mtris[ref_tri.empty() ? id : ref_tri[id]]);
        encode_tree(wide, mr.node_base, mr.tri_base, blas_nodes, blas_boxes);
        if (mr.node_capacity > mr.node_count) { // room for a device-side rebuild of this dynamic mesh (lbvh.h): unreachable empty nodes
            rptr::Wide4 pad_src;
            (void)pad_src;
            RptrBvh4Node empty;
            memset(&empty, 0, sizeof(empty));
            for (int k = 0; k < 4; ++k) empty.child[k] = RPTR_BVH4_EMPTY;
            blas_nodes.resize((size_t)mr.node_base + mr.node_capacity, empty);
            blas_boxes.resize((size_t)mr.node_base + mr.node_capacity, std::array<float, 6>{0, 0, 0, 0, 0, 0});
        }
    }
    // ---- top level over instance bounds (1 instance record per leaf). Partial re-braiding: when many instances overlap
    // (a forest), one box per instance makes rays enter instance after instance just to leave them at the first nodes.
    // An instance is then represented by up to `braid` records that share transform and ids but start at different
    // sub-roots of its bottom-level tree (the cut is opened largest box first, only through nodes whose children are all
    // inner nodes), each with the world box of its own subtree.
    int braid = s->num_instances >= 16 ? 4 : 1;
    if (opt.v[OPT_REBRAID] > 0) braid = (int)opt.v[OPT_REBRAID];
    if (flatten) braid = 1; // (PARTIAL: the dynamic meshes' instances keep one record each: flat_bias counts on it)
    std::vector<rptr::BuildPrim> iprims;
    std::vector<RptrBvhInstance> insts;
    iprims.reserve((size_t)s->num_instances * braid);
    insts.reserve((size_t)s->num_instances * braid);
    std::vector<std::vector<int>> mesh_cut(B.meshes.size()); // per mesh: the sub-roots (absolute BLAS node indices, before relocation)
    for (size_t m = 0; m < B.meshes.size(); ++m) {
        std::vector<int> cut{B.meshes[m].node_base};
        auto area = [&](int n) {
            const std::array<float, 6> &b = blas_boxes[n];
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            return dx * dy + dy * dz + dz * dx;
        };
        // (a mesh the BVH policy may rebuild keeps its one root: a device-side rebuild gives it a new topology, and sub-roots named by
        // instance records would then point into the middle of another tree; refits keep the topology)
        while ((int)cut.size() < braid && !B.meshes[m].rebuildable) {
            int pick = -1;
            float best = -1.0f;
            for (size_t i = 0; i < cut.size(); ++i) {
                const RptrBvh4Node &nd = blas_nodes[cut[i]];
                int inner = 0, other = 0;
                for (int k = 0; k < 4; ++k) {
                    if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                    (nd.child[k] >= 0 ? inner : other)++;
                }
                if (inner < 2 || other > 0 || (int)cut.size() - 1 + inner > braid) continue; // leaves below it / would overshoot
                const float a = area(cut[i]);
                if (a > best) {
                    best = a;
                    pick = (int)i;
                }
            }
            if (pick < 0) break;
            const RptrBvh4Node nd = blas_nodes[cut[pick]];
            cut.erase(cut.begin() + pick);
            for (int k = 0; k < 4; ++k)
                if (nd.child[k] >= 0) cut.push_back(nd.child[k]);
        }
        mesh_cut[m] = cut;
    }
    std::vector<RptrBvhInstance> own_records; // flattened scene: the scene's instance records, behind the one the top level uses
    if (flatten) {
        RptrBvhInstance bi;
        memset(&bi, 0, sizeof(bi));
        const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(bi.object_to_world, identity, 48);
        memcpy(bi.world_to_object, identity, 48);
        bi.blas_root = 0; // relocated below
        bi.instance_id = -1;
        // (FLAT promises ONE top-level record -- queries start inside it, csrc/dtraverse.h SINGLE, oracle/obvh.h --; the flat tree of a
        // partially flattened scene is an ordinary identity instance of the top level whose triangles name their own records)
        bi.flags = partial ? 0 : RPTR_BVH_INSTANCE_FLAT;
        insts.push_back(bi);
        rptr::BuildPrim bp;
        const std::array<float, 6> &mb = blas_boxes[0];
        for (int k = 0; k < 3; ++k) {
            bp.lo[k] = mb[k];
            bp.hi[k] = mb[3 + k];
        }
        iprims.push_back(bp);
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            RptrBvhInstance r;
            memset(&r, 0, sizeof(r));
            memcpy(r.object_to_world, in.transform, 48);
            invert_affine(in.transform, r.world_to_object);
            r.blas_root = -1;
            r.geometry_base = pmesh_base[in.parameterized_mesh];
            r.instance_id = (int)i;
            own_records.push_back(r);
        }
    }
    // the shading records of a mesh's triangles carry the material ids of the FIRST parameterized mesh that uses the mesh (set_scene builds
    // them): instances of any other one resolve theirs through their geometry records
    std::vector<int> first_pmesh_of_mesh(s->num_meshes, -1);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p)
        if (first_pmesh_of_mesh[s->parameterized_meshes[p].mesh] < 0) first_pmesh_of_mesh[s->parameterized_meshes[p].mesh] = (int)p;
    for (uint32_t i = 0; i < s->num_instances && (!flatten || partial); ++i) {
        const RptrInstanceDesc &in = s->instances[i];
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
        if (partial && !s->meshes[pm.mesh].dynamic) continue;
        RptrBvhInstance bi;
        memset(&bi, 0, sizeof(bi));
        memcpy(bi.object_to_world, in.transform, 48);
        invert_affine(in.transform, bi.world_to_object);
        bi.geometry_base = pmesh_base[in.parameterized_mesh];
        bi.instance_id = (int)i;
        if (first_pmesh_of_mesh[pm.mesh] != (int)in.parameterized_mesh) bi.flags |= RPTR_BVH_INSTANCE_OWN_MATERIALS;
        for (int sub : mesh_cut[pm.mesh]) {
            bi.blas_root = sub; // relocated below
            insts.push_back(bi);
            const std::array<float, 6> &mb = blas_boxes[sub]; // exact bounds of the subtree (= the mesh for the root)
            rptr::BuildPrim bp;
            for (int k = 0; k < 3; ++k) {
                bp.lo[k] = INFINITY;
                bp.hi[k] = -INFINITY;
            }
            for (int c = 0; c < 8; ++c) {
                const float p[3] = {c & 1 ? mb[3] : mb[0], c & 2 ? mb[4] : mb[1], c & 4 ? mb[5] : mb[2]};
                const float *M = in.transform;
                for (int r = 0; r < 3; ++r) {
                    const float w = ((M[4 * r] * p[0] + M[4 * r + 1] * p[1]) + M[4 * r + 2] * p[2]) + M[4 * r + 3];
                    bp.lo[r] = std::fmin(bp.lo[r], w);
                    bp.hi[r] = std::fmax(bp.hi[r], w);
                }
            }
            iprims.push_back(bp);
        }
    }
    rptr::BuiltTree tlas;
    rptr::build_bvh2(iprims.data(), (uint32_t)iprims.size(), 1, 24, 1, tlas);
    rptr::Wide4Tree tlas_wide;
    // (the top level keeps the greedy rule: over the heavily overlapping instance boxes of a forest the area-optimal collapse needs 38.5 node
    // visits per ray where the greedy one needs 36.8 -- there the area of a box says little about what a ray does inside it;
    // RPTR_TLAS_COLLAPSE=optimal to try)
    const int tlas_rule = (int)opt.v[OPT_TLAS_COLLAPSE]; // (rptr::COLLAPSE_*: 0 greedy)
    rptr::collapse_bvh4(tlas, tlas_wide, tlas_rule);
    for (int k = 0; k < 3; ++k) {
        B.scene_lo[k] = std::isfinite(tlas.lo[k]) ? tlas.lo[k] : 0.0f;
        B.scene_hi[k] = std::isfinite(tlas.hi[k]) ? tlas.hi[k] : 1.0f;
    }
    const int reloc = (int)tlas_wide.nodes.size();
    B.num_tlas_nodes = reloc;
    B.nodes.clear();
    B.node_box.clear();
    encode_tree(tlas_wide, 0, 0, B.nodes, B.node_box); // TLAS leaf 'first' already indexes the reordered instance array
    for (size_t i = 0; i < blas_nodes.size(); ++i) {
        RptrBvh4Node nd = blas_nodes[i];
        for (int k = 0; k < 4; ++k)
            if (nd.child[k] >= 0) nd.child[k] += reloc;
        B.nodes.push_back(nd);
        B.node_box.push_back(blas_boxes[i]);
    }
    B.mesh_root.assign(B.meshes.size(), -1);
    for (size_t m = 0; m < B.meshes.size(); ++m) {
        B.meshes[m].node_base += reloc;
        B.mesh_root[m] = B.meshes[m].node_base;
    }
    B.insts.resize(insts.size());
    for (size_t k = 0; k < insts.size(); ++k) {
        B.insts[k] = insts[tlas.order[k]];
        B.insts[k].blas_root += reloc;
    }
    B.num_tlas_insts = (int)B.insts.size();
    B.insts.insert(B.insts.end(), own_records.begin(), own_records.end());
    // ---- the traversal stack must hold the worst case of this tree: per node (children - 1) siblings plus whatever
    // its deepest child needs; + the exit marker, + the instance-exit sentinel between the two levels
    {
        const size_t nn = B.nodes.size();
        std::vector<int> need(nn, 0);
        for (int64_t i = (int64_t)nn - 1; i >= 0; --i) { // children sit behind their parents (breadth-first order per tree)
            const RptrBvh4Node &nd = B.nodes[i];
            int nchild = 0, deepest = 0;
            for (int k = 0; k < 4; ++k) {
                if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                ++nchild;
                if (nd.child[k] >= 0) deepest = std::max(deepest, need[nd.child[k]]);
            }
            need[i] = std::max(0, nchild - 1) + deepest;
        }
        int blas_need = 0;
        for (size_t m = 0; m < B.meshes.size(); ++m) blas_need = std::max(blas_need, need[B.mesh_root[m]]);
        const int total = 1 + need[0] + 1 + blas_need;
        B.stack_need = total;
    }
}

static int drain(rptr_hip *h);
static int build_shade_records(rptr_hip *h, SceneCopy &sc, int only_mesh, hipStream_t st);
}

