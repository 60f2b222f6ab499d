// TEST INFRASTRUCTURE -- CPU oracle: acceleration structure + ray queries.
//
// The reference delegates BVH build, traversal and the ray/triangle test to the
// Vulkan driver (vulkan/pt_megakernel.glsl:440-475, :216-272,
// vulkan/rt_intersect.comp:46-51; SURVEY 8(a4/a13/a19)): there is NO reference
// source for this part -> "parity unpinned". What the oracle fixes instead is a
// *canonical* query semantics that any correct implementation must reproduce
// bit for bit, independent of the tree:
//   - Moeller-Trumbore in the object space of the mesh, on the dequantised
//     float positions (the reference feeds dequantised floats to the BLAS
//     build, render_vulkan.cpp:698-711), operations exactly as written in
//     mt_intersect() below (explicit fmaf, nothing else contracted);
//   - a hit is accepted for t_min < t < t_max; barycentrics (u,v) weight
//     vertex 1 and 2 (rendering/rt/hit.glsl:70);
//   - closest hit = smallest t; equal t is resolved towards the smallest
//     (instance, geometry, primitive) triple, so the answer does not depend on
//     traversal order;
//   - shadow query = "is there any accepted hit".
// Three implementations live here: brute force (ground truth for small
// scenes), traversal of an oracle-built BVH, and traversal of a BVH exported by
// the product (rptr_hip_export_bvh) with node/triangle visit counting -- the
// latter defines the algorithmic-bytes figure of the roofline.
#pragma once
#include "../include/rptr_bvh.h"
#include "../include/rptr_hip.h"
#include "oshade.h"
#include <algorithm>
#include <array>
#include <cfloat>
#include <cstdlib>
#include <vector>

namespace orc {

struct Ray {
    vec3 o, d;
    float tmin, tmax;
};
struct Hit {
    float t, u, v;
    int inst;  // index into the instance array (-1: miss)
    int geom;  // geometry index inside the mesh
    int prim;  // primitive index inside the geometry
    vec3 lo, ld; // object-space ray of the committed hit
};
struct TraceCounters {
    uint64_t nodes = 0, tris = 0;
};
// optional per-node visit histogram (diagnostics: which part of the tree is hot)
// ORC_BASELINE (oracle/Makefile libcpu_baseline.so): the same restatement built as the CPU baseline of bench.py -- -O3 -march=native, no
// diagnostics: the node histogram, the ray log and the visit counters are compiled out (constant null pointers / false).
#ifdef ORC_BASELINE
static uint32_t *const g_node_hist = nullptr;
#else
static uint32_t *g_node_hist = nullptr;
#endif

static inline bool hit_key_less(int inst, int geom, int prim, const Hit &h) {
    if (h.inst < 0) return true;
    if (inst != h.inst) return inst < h.inst;
    if (geom != h.geom) return geom < h.geom;
    return prim < h.prim;
}

// canonical ray/triangle test (see header): Moeller-Trumbore with explicit fused
// multiply-adds (one rounding per dot/cross term, identical on host and device)
// and a division-free inside test; the reciprocal of the determinant is only
// formed for triangles whose plane hit lies inside. Returns true and t,u,v; the
// caller applies the ray interval.
static inline float dot_fma(const vec3 a, const vec3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline vec3 cross_fma(const vec3 a, const vec3 b) {
    return vec3(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y)));
}
static inline bool mt_intersect(const vec3 o, const vec3 d, const vec3 v0, const vec3 e1, const vec3 e2, float &t, float &u, float &v) {
    const vec3 p = cross_fma(d, e2);
    const float det = dot_fma(e1, p);
    const vec3 tv = o - v0;
    const float un = dot_fma(tv, p);
    const vec3 q = cross_fma(tv, e1);
    const float vn = dot_fma(d, q);
    const float ad = fabsf(det);
    const bool neg = std::signbit(det);
    const float us = neg ? -un : un, vs = neg ? -vn : vn;
    if (!(us >= 0.0f && vs >= 0.0f && us + vs <= ad && ad > 0.0f)) return false;
    const float inv = 1.0f / det;
    t = dot_fma(e2, q) * inv;
    u = un * inv;
    v = vn * inv;
    return true;
}

static inline float safe_rcp(float x) { return fabsf(x) >= 1e-30f ? 1.0f / x : copysignf(1e30f, x); }

// slab test shared by every traversal (and restated by the HIP kernels):
// returns entry distance in tnear; hit iff tnear <= tfar * (1 + 2^-19).
static inline bool slab(const float lo[3], const float hi[3], const vec3 o, const vec3 id, float tmin, float tmax, float &tnear) {
    float t0x = (lo[0] - o.x) * id.x, t1x = (hi[0] - o.x) * id.x;
    float t0y = (lo[1] - o.y) * id.y, t1y = (hi[1] - o.y) * id.y;
    float t0z = (lo[2] - o.z) * id.z, t1z = (hi[2] - o.z) * id.z;
    tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tnear <= tfar * 1.0000019f;
}

static inline vec3 xform_point(const float m[12], vec3 p) {
    return vec3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
                ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline vec3 xform_dir(const float m[12], vec3 d) {
    return vec3((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z, (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
}

// inverse of a row-major 3x4 affine transform, cofactors in double, rounded once.
static inline void invert_affine(const float m[12], float out[12]) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
    double det = a * A + b * B + c * C;
    double id = 1.0 / det;
    double r[9] = {A * id, -(b * i - c * h) * id, (b * f - c * e) * id, B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                   C * id, -(a * h - b * g) * id, (a * e - b * d) * id};
    double tx = m[3], ty = m[7], tz = m[11];
    for (int k = 0; k < 3; ++k) {
        out[4 * k + 0] = (float)r[3 * k + 0];
        out[4 * k + 1] = (float)r[3 * k + 1];
        out[4 * k + 2] = (float)r[3 * k + 2];
        out[4 * k + 3] = (float)(-(r[3 * k + 0] * tx + r[3 * k + 1] * ty + r[3 * k + 2] * tz));
    }
}

// ------------------------------------------------------------------ scene view
struct GeomRecord { // ≙ RenderMeshParams per (parameterized mesh, geometry), render_vulkan.cpp:2748-2850
    const RptrGeometryDesc *g;
    int material_id;         // >=0 or -1-offset
    const uint8_t *mat_ids;  // per-triangle ids of THIS geometry (already offset by primOffset) or NULL
};
struct SceneView {
    const RptrSceneDesc *desc = nullptr;
    std::vector<GeomRecord> geoms;         // instanced_geometry[]
    std::vector<int> pmesh_geom_base;      // render_mesh_base_offset per parameterized mesh
    // float positions of dynamic-mesh geometries (Geometry.dynamic_vertices, pt_megakernel.glsl:526-529),
    // indexed by global geometry; empty = read the quantised stream
    std::vector<std::vector<float>> dyn_pos;
    void init(const RptrSceneDesc *s) {
        desc = s;
        dyn_pos.assign(s->num_geometries, {});
        geoms.clear();
        pmesh_geom_base.clear();
        for (uint32_t pm = 0; pm < s->num_parameterized_meshes; ++pm) {
            const auto &p = s->parameterized_meshes[pm];
            const auto &mesh = s->meshes[p.mesh];
            pmesh_geom_base.push_back((int)geoms.size());
            size_t prim_offset = 0;
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                const RptrGeometryDesc *g = &s->geometries[mesh.first_geometry + j];
                GeomRecord r;
                r.g = g;
                if (p.tri_material_ids) {
                    r.material_id = -1 - p.material_offsets[j];
                    r.mat_ids = p.tri_material_ids + prim_offset;
                } else {
                    r.material_id = p.material_offsets[j];
                    r.mat_ids = nullptr;
                }
                geoms.push_back(r);
                prim_offset += g->num_tris;
            }
        }
    }
};

static inline void geom_tri(const SceneView &view, const RptrGeometryDesc &g, uint32_t prim, vec3 &v0, vec3 &v1, vec3 &v2) {
    const std::vector<float> &dyn = view.dyn_pos[&g - view.desc->geometries];
    if (!dyn.empty()) {
        const float *p = &dyn[9 * (size_t)prim];
        v0 = vec3(p[0], p[1], p[2]);
        v1 = vec3(p[3], p[4], p[5]);
        v2 = vec3(p[6], p[7], p[8]);
        return;
    }
    vec3 sc(g.quantized_scaling[0], g.quantized_scaling[1], g.quantized_scaling[2]);
    vec3 of(g.quantized_offset[0], g.quantized_offset[1], g.quantized_offset[2]);
    v0 = dequantize_position(g.qpos[3 * prim + 0], sc, of);
    v1 = dequantize_position(g.qpos[3 * prim + 1], sc, of);
    v2 = dequantize_position(g.qpos[3 * prim + 2], sc, of);
}

// ------------------------------------------------------------------ BVH container (own build or imported)
struct Bvh {
    std::vector<RptrBvhNode> nodes;   // the oracle's own tree: 2-wide, float boxes
    std::vector<RptrBvh4Node> nodes4; // a tree imported from the device (rptr_hip_export_bvh): 4-wide, 8-bit boxes
    std::vector<RptrBvhTri> tris;
    std::vector<RptrBvhInstance> insts;
};

struct BuildRef {
    float lo[3], hi[3], c[3];
    uint32_t id;
};
struct TmpNode {
    float lo[3], hi[3];
    int left, right;   // children (TmpNode indices) or -1
    uint32_t first, count; // leaf range in the reordered ref array
};

static inline void box_init(float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
}
static inline void box_grow(float lo[3], float hi[3], const float l2[3], const float h2[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], l2[k]); hi[k] = fmaxf(hi[k], h2[k]); }
}
static inline float box_area(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

// binned SAH (16 bins), the oracle's own builder; quality only matters for the
// CPU baseline's speed, never for results.
static int build_rec(std::vector<BuildRef> &refs, std::vector<TmpNode> &out, uint32_t begin, uint32_t end, uint32_t max_leaf) {
    TmpNode node;
    box_init(node.lo, node.hi);
    float clo[3], chi[3];
    box_init(clo, chi);
    for (uint32_t i = begin; i < end; ++i) {
        box_grow(node.lo, node.hi, refs[i].lo, refs[i].hi);
        box_grow(clo, chi, refs[i].c, refs[i].c);
    }
    node.left = node.right = -1;
    node.first = begin;
    node.count = end - begin;
    int self = (int)out.size();
    out.push_back(node);
    if (end - begin <= max_leaf) return self;
    const int NB = 16;
    int best_axis = -1, best_bin = -1;
    float best_cost = INFINITY;
    for (int ax = 0; ax < 3; ++ax) {
        float ext = chi[ax] - clo[ax];
        if (!(ext > 0)) continue;
        float blo[NB][3], bhi[NB][3];
        uint32_t bcnt[NB];
        for (int b = 0; b < NB; ++b) { box_init(blo[b], bhi[b]); bcnt[b] = 0; }
        float scale = NB / ext;
        for (uint32_t i = begin; i < end; ++i) {
            int b = (int)((refs[i].c[ax] - clo[ax]) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            box_grow(blo[b], bhi[b], refs[i].lo, refs[i].hi);
            bcnt[b]++;
        }
        float la[NB], ra[NB];
        uint32_t lc[NB], rc[NB];
        float alo[3], ahi[3];
        box_init(alo, ahi);
        uint32_t cnt = 0;
        for (int b = 0; b < NB; ++b) { box_grow(alo, ahi, blo[b], bhi[b]); cnt += bcnt[b]; la[b] = box_area(alo, ahi); lc[b] = cnt; }
        box_init(alo, ahi);
        cnt = 0;
        for (int b = NB - 1; b >= 0; --b) { box_grow(alo, ahi, blo[b], bhi[b]); cnt += bcnt[b]; ra[b] = box_area(alo, ahi); rc[b] = cnt; }
        for (int b = 0; b < NB - 1; ++b) {
            if (lc[b] == 0 || rc[b + 1] == 0) continue;
            float cost = la[b] * lc[b] + ra[b + 1] * rc[b + 1];
            if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = b; }
        }
    }
    uint32_t mid;
    if (best_axis < 0) {
        mid = (begin + end) / 2; // all centroids coincide: split by index
    } else {
        float ext = chi[best_axis] - clo[best_axis];
        float scale = NB / ext;
        auto it = std::partition(refs.begin() + begin, refs.begin() + end, [&](const BuildRef &r) {
            int b = (int)((r.c[best_axis] - clo[best_axis]) * scale);
            b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
            return b <= best_bin;
        });
        mid = (uint32_t)(it - refs.begin());
        if (mid == begin || mid == end) mid = (begin + end) / 2;
    }
    int l = build_rec(refs, out, begin, mid, max_leaf);
    int r = build_rec(refs, out, mid, end, max_leaf);
    out[self].left = l;
    out[self].right = r;
    return self;
}

// flatten TmpNode tree into the 2-children-per-node layout. leaf_base: offset
// added to leaf 'first'. Returns absolute index of the root node appended to
// `nodes`. A tree that is a single leaf gets a root whose child1 is empty.
static int flatten(const std::vector<TmpNode> &tmp, std::vector<RptrBvhNode> &nodes, uint32_t leaf_base) {
    auto is_leaf = [&](int i) { return tmp[i].left < 0; };
    struct Item { int tmp_idx; int out_idx; };
    int root_out = (int)nodes.size();
    nodes.push_back(RptrBvhNode());
    auto set_child = [&](RptrBvhNode &n, int which, int tmp_idx, int out_idx) {
        float *lo = which ? n.lo1 : n.lo0, *hi = which ? n.hi1 : n.hi0;
        for (int k = 0; k < 3; ++k) { lo[k] = tmp[tmp_idx].lo[k]; hi[k] = tmp[tmp_idx].hi[k]; }
        if (is_leaf(tmp_idx)) {
            (which ? n.child1 : n.child0) = RPTR_BVH_LEAF(leaf_base + tmp[tmp_idx].first, tmp[tmp_idx].count);
            (which ? n.cnt1 : n.cnt0) = (int32_t)tmp[tmp_idx].count;
        } else {
            (which ? n.child1 : n.child0) = out_idx;
            (which ? n.cnt1 : n.cnt0) = 0;
        }
    };
    if (is_leaf(0)) {
        RptrBvhNode n;
        set_child(n, 0, 0, -1);
        box_init(n.lo1, n.hi1);
        n.child1 = RPTR_BVH_LEAF(0, 0);
        n.cnt1 = 0;
        nodes[root_out] = n;
        return root_out;
    }
    std::vector<Item> stack;
    stack.push_back({0, root_out});
    while (!stack.empty()) {
        Item it = stack.back();
        stack.pop_back();
        int l = tmp[it.tmp_idx].left, r = tmp[it.tmp_idx].right;
        int lo_idx = -1, ro_idx = -1;
        if (!is_leaf(l)) { lo_idx = (int)nodes.size(); nodes.push_back(RptrBvhNode()); }
        if (!is_leaf(r)) { ro_idx = (int)nodes.size(); nodes.push_back(RptrBvhNode()); }
        RptrBvhNode n;
        set_child(n, 0, l, lo_idx);
        set_child(n, 1, r, ro_idx);
        nodes[it.out_idx] = n;
        if (!is_leaf(r)) stack.push_back({r, ro_idx});
        if (!is_leaf(l)) stack.push_back({l, lo_idx});
    }
    return root_out;
}

static void build_bvh(const SceneView &sv, Bvh &bvh) {
    const RptrSceneDesc *s = sv.desc;
    bvh.nodes.clear();
    bvh.tris.clear();
    bvh.insts.clear();
    // TLAS placeholder is filled last but must sit at node 0: build BLAS into a
    // temporary array first, then relocate.
    std::vector<RptrBvhNode> blas_nodes;
    std::vector<int> mesh_root(s->num_meshes, -1);
    std::vector<std::array<float, 6>> mesh_box(s->num_meshes);
    // RPTR_BVH_TRI_ALPHA (include/rptr_bvh.h): the triangle has a material without BASE_MATERIAL_NOALPHA in some
    // parameterized mesh of its mesh
    std::vector<std::vector<uint8_t>> tri_alpha(s->num_meshes);
    for (uint32_t pm = 0; pm < s->num_parameterized_meshes; ++pm) {
        const auto &p = s->parameterized_meshes[pm];
        const auto &mesh = s->meshes[p.mesh];
        size_t at = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j)
            for (uint32_t t = 0; t < s->geometries[mesh.first_geometry + j].num_tris; ++t, ++at) {
                if (tri_alpha[p.mesh].size() <= at) tri_alpha[p.mesh].resize(at + 1, 0);
                const long mid = (long)p.material_offsets[j] + (p.tri_material_ids ? (long)p.tri_material_ids[at] : 0);
                if (mid >= 0 && mid < (long)s->num_materials && !(s->materials[mid].flags & RPTR_BASE_MATERIAL_NOALPHA)) tri_alpha[p.mesh][at] = 1;
            }
    }
    for (uint32_t m = 0; m < s->num_meshes; ++m) {
        const auto &mesh = s->meshes[m];
        std::vector<BuildRef> refs;
        std::vector<RptrBvhTri> mtris;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const auto &g = s->geometries[mesh.first_geometry + j];
            for (uint32_t p = 0; p < g.num_tris; ++p) {
                vec3 v0, v1, v2;
                geom_tri(sv, g, p, v0, v1, v2);
                RptrBvhTri t;
                vec3 e1 = v1 - v0, e2 = v2 - v0;
                t.v0[0] = v0.x; t.v0[1] = v0.y; t.v0[2] = v0.z;
                t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
                t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
                t.prim = p; t.geom = j;
                t.flags = (mtris.size() < tri_alpha[m].size() && tri_alpha[m][mtris.size()]) ? RPTR_BVH_TRI_ALPHA : 0u;
                BuildRef r;
                for (int k = 0; k < 3; ++k) {
                    r.lo[k] = fminf(v0[k], fminf(v1[k], v2[k]));
                    r.hi[k] = fmaxf(v0[k], fmaxf(v1[k], v2[k]));
                    r.c[k] = 0.5f * (r.lo[k] + r.hi[k]);
                }
                r.id = (uint32_t)mtris.size();
                mtris.push_back(t);
                refs.push_back(r);
            }
        }
        std::vector<TmpNode> tmp;
        if (refs.empty()) {
            TmpNode n; box_init(n.lo, n.hi); n.left = n.right = -1; n.first = 0; n.count = 0; tmp.push_back(n);
        } else {
            tmp.reserve(refs.size() * 2);
            build_rec(refs, tmp, 0, (uint32_t)refs.size(), RPTR_BVH_MAX_LEAF_TRIS);
        }
        uint32_t tri_base = (uint32_t)bvh.tris.size();
        for (auto &r : refs) bvh.tris.push_back(mtris[r.id]);
        mesh_root[m] = flatten(tmp, blas_nodes, tri_base);
        for (int k = 0; k < 3; ++k) { mesh_box[m][k] = tmp[0].lo[k]; mesh_box[m][3 + k] = tmp[0].hi[k]; }
    }
    // instances + TLAS (1 instance per leaf)
    std::vector<BuildRef> irefs;
    for (uint32_t i = 0; i < s->num_instances; ++i) {
        const auto &in = s->instances[i];
        const auto &pm = s->parameterized_meshes[in.parameterized_mesh];
        RptrBvhInstance bi;
        memset(&bi, 0, sizeof(bi));
        memcpy(bi.object_to_world, in.transform, sizeof(float) * 12);
        invert_affine(in.transform, bi.world_to_object);
        bi.blas_root = mesh_root[pm.mesh]; // relocated below
        bi.geometry_base = sv.pmesh_geom_base[in.parameterized_mesh];
        bi.instance_id = (int)i;
        bvh.insts.push_back(bi);
        BuildRef r;
        box_init(r.lo, r.hi);
        const auto &mb = mesh_box[pm.mesh];
        for (int c = 0; c < 8; ++c) {
            vec3 p(c & 1 ? mb[3] : mb[0], c & 2 ? mb[4] : mb[1], c & 4 ? mb[5] : mb[2]);
            vec3 w = xform_point(in.transform, p);
            float wl[3] = {w.x, w.y, w.z};
            box_grow(r.lo, r.hi, wl, wl);
        }
        for (int k = 0; k < 3; ++k) r.c[k] = 0.5f * (r.lo[k] + r.hi[k]);
        r.id = i;
        irefs.push_back(r);
    }
    std::vector<TmpNode> ttmp;
    if (irefs.empty()) {
        TmpNode n; box_init(n.lo, n.hi); n.left = n.right = -1; n.first = 0; n.count = 0; ttmp.push_back(n);
    } else
        build_rec(irefs, ttmp, 0, (uint32_t)irefs.size(), 1);
    // reorder instance records to leaf order
    std::vector<RptrBvhInstance> ordered;
    for (auto &r : irefs) ordered.push_back(bvh.insts[r.id]);
    bvh.insts = ordered;
    flatten(ttmp, bvh.nodes, 0);
    int reloc = (int)bvh.nodes.size();
    for (auto n : blas_nodes) {
        if (n.child0 >= 0) n.child0 += reloc;
        if (n.child1 >= 0) n.child1 += reloc;
        bvh.nodes.push_back(n);
    }
    for (auto &bi : bvh.insts) bi.blas_root += reloc;
}

// ------------------------------------------------------------------ traversal
// One algorithm, closest or any-hit; it defines the canonical visit order (and
// with it the node / triangle counts of the roofline model):
//   * work items are inner nodes (>= 0), leaves (encoded <= -2) and the
//     instance-exit sentinel; the current item is processed, then the next one
//     is popped from the stack;
//   * inner node: fetch, slab-test both children against [t_min, best_t]; if both
//     are hit, the farther one is pushed and the nearer one becomes current
//     (tie -> child 0 is nearer); if one is hit it becomes current;
//   * BLAS leaf: every triangle of the leaf is tested; TLAS leaf (one instance):
//     the ray is transformed into object space, the sentinel is pushed and the
//     instance's root becomes current.
static inline void decode_leaf(int enc, int &first, int &count) {
    first = RPTR_BVH_LEAF_FIRST(enc);
    count = RPTR_BVH_LEAF_COUNT(enc);
}
static bool g_no_single_instance = getenv("RPTR_NO_SINGLE_INSTANCE") != nullptr; // same switch as the device library
static const bool g_sort_by_distance = getenv("ORC_SORT_BY_DISTANCE") != nullptr; // diagnostic (tools/order_probe.py): a full sort by entry distance instead of the three comparisons
static const int g_order_key = getenv("ORC_ORDER_KEY") ? atoi(getenv("ORC_ORDER_KEY")) : 0; // diagnostic: 1 = box midpoint along the ray, 2 = unclamped entry (all queries), 3 = exit, 4 = entry then exit, 5 / 6 = ties to the smaller / larger box, 7 = clamped entry for every query (the order of rounds 1-3a)
static unsigned long long *g_dead_visits = nullptr; // diagnostic: node visits in which no child box was hit
// The reference's any-hit stage (vulkan/pt_megakernel.glsl:153-212, generate_candidate_hit): called for every hit of a
// triangle flagged RPTR_BVH_TRI_ALPHA that the query would otherwise accept, in the canonical order of this traversal;
// true = the candidate is ignored and the traversal goes on. The reference leaves the order in which candidates turn up
// to the driver ("parity unpinned"); oracle and device agree on the order below.
struct AlphaTest {
    virtual bool reject(const RptrBvhInstance &inst, const RptrBvhTri &tri, float t, float u, float v) = 0;
    virtual ~AlphaTest() {}
};
template <bool ANY>
static bool traverse4(const Bvh &bvh, const Ray &ray, Hit &best, TraceCounters *cnt, AlphaTest *alpha);
template <bool ANY>
static bool traverse2(const Bvh &bvh, const Ray &ray, Hit &best, TraceCounters *cnt, AlphaTest *alpha);
template <bool ANY>
static bool traverse(const Bvh &bvh, const Ray &ray, Hit &best, TraceCounters *cnt, AlphaTest *alpha = nullptr) {
    return bvh.nodes4.empty() ? traverse2<ANY>(bvh, ray, best, cnt, alpha) : traverse4<ANY>(bvh, ray, best, cnt, alpha);
}
template <bool ANY>
static bool traverse2(const Bvh &bvh, const Ray &ray, Hit &best, TraceCounters *cnt, AlphaTest *alpha) {
    best.t = ray.tmax;
    best.inst = -1;
    best.u = best.v = 0;
    best.geom = best.prim = -1;
    const RptrBvhNode *nodes = bvh.nodes.data();
    int stack[RPTR_BVH_STACK_DEPTH];
    int sp = 0;
    const int SENTINEL = INT32_MIN;
    vec3 o = ray.o, d = ray.d;
    vec3 id(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const RptrBvhInstance *cur_inst = nullptr;
    int cur = 0;
    for (;;) {
        bool pop = false;
        if (cur >= 0) {
            const RptrBvhNode &n = nodes[cur];
            if (cnt) cnt->nodes++;
            if (g_node_hist) __atomic_fetch_add(&g_node_hist[cur], 1u, __ATOMIC_RELAXED);
            float tn0, tn1;
            const bool h0 = slab(n.lo0, n.hi0, o, id, ray.tmin, best.t, tn0);
            const bool h1 = slab(n.lo1, n.hi1, o, id, ray.tmin, best.t, tn1);
            const int c0 = n.child0, c1 = n.child1;
            if (h0 && h1) {
                const bool near1 = tn1 < tn0;
                stack[sp++] = near1 ? c0 : c1;
                cur = near1 ? c1 : c0;
            } else if (h0)
                cur = c0;
            else if (h1)
                cur = c1;
            else
                pop = true;
        } else if (cur == SENTINEL) {
            cur_inst = nullptr;
            o = ray.o;
            d = ray.d;
            id = vec3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
            pop = true;
        } else {
            int first, count;
            decode_leaf(cur, first, count);
            if (cur_inst == nullptr) {
                if (count > 0) {
                    cur_inst = &bvh.insts[first];
                    if (cnt) cnt->nodes += 2; // 128-byte instance record = 2 node-sized fetches
                    o = xform_point(cur_inst->world_to_object, ray.o);
                    d = xform_dir(cur_inst->world_to_object, ray.d);
                    id = vec3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
                    stack[sp++] = SENTINEL;
                    cur = cur_inst->blas_root;
                } else
                    pop = true;
            } else {
                for (int k = 0; k < count; ++k) {
                    const RptrBvhTri &tr = bvh.tris[first + k];
                    if (cnt) cnt->tris++;
                    float t, u, v;
                    if (!mt_intersect(o, d, vec3(tr.v0[0], tr.v0[1], tr.v0[2]), vec3(tr.e1[0], tr.e1[1], tr.e1[2]),
                                      vec3(tr.e2[0], tr.e2[1], tr.e2[2]), t, u, v))
                        continue;
                    if (!(t > ray.tmin)) continue;
                    const int ii = cur_inst->instance_id;
                    const bool accept = (t < best.t) || (t == best.t && best.inst >= 0 && hit_key_less(ii, (int)tr.geom, (int)tr.prim, best));
                    if (!accept) continue;
                    if (alpha && (tr.flags & RPTR_BVH_TRI_ALPHA) && alpha->reject(*cur_inst, tr, t, u, v)) continue;
                    best.t = t; best.u = u; best.v = v;
                    best.inst = ii; best.geom = (int)tr.geom; best.prim = (int)tr.prim;
                    best.lo = o; best.ld = d;
                    if (ANY) return true;
                }
                pop = true;
            }
        }
        if (pop) {
            if (sp == 0) return best.inst >= 0;
            cur = stack[--sp];
        }
    }
}

// The device's tree (include/rptr_bvh.h RptrBvh4Node), walked in the device's canonical order
// (csrc/dtraverse.h header): per node the four child boxes are tested on the node's 8-bit grid with
// t = fma(q, A, B), A = step/d, B = (origin - o)/d; hit children are ordered by three comparisons of their entry
// distances (inside slot pairs (0,1) and (2,3), then pair against pair); the first is visited next, the others are
// pushed so that the first of the rest pops next. Leaves, instances and the triangle test are those of traverse2.
template <bool ANY>
static bool traverse4(const Bvh &bvh, const Ray &ray, Hit &best, TraceCounters *cnt, AlphaTest *alpha) {
    best.t = ray.tmax;
    best.inst = -1;
    best.u = best.v = 0;
    best.geom = best.prim = -1;
    const RptrBvh4Node *nodes = bvh.nodes4.data();
    int stack[4 * RPTR_BVH_STACK_DEPTH];
    float stack_tn[4 * RPTR_BVH_STACK_DEPTH];
    int sp = 0;
    const int SENTINEL = INT32_MIN;
    vec3 o = ray.o, d = ray.d;
    vec3 id(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
    const RptrBvhInstance *cur_inst = nullptr;
    int cur = 0;
    // (a flattened scene keeps the scene's own instance records behind the one its top level refers to: rptr_bvh.h)
    if ((bvh.insts.size() == 1 || (!bvh.insts.empty() && (bvh.insts[0].flags & RPTR_BVH_INSTANCE_FLAT))) && !g_no_single_instance) {
        // the device's shortcut for scenes with one instance record (csrc/dtraverse.h): start inside the instance
        cur_inst = &bvh.insts[0];
        if (cnt) cnt->nodes++; // the first 64 bytes of the instance record
        o = xform_point(cur_inst->world_to_object, ray.o);
        d = xform_dir(cur_inst->world_to_object, ray.d);
        id = vec3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
        cur = cur_inst->blas_root;
    }
    for (;;) {
        bool pop = false;
        if (cur >= 0) {
            const RptrBvh4Node &n = nodes[cur];
            if (cnt) cnt->nodes++;
            if (g_node_hist) __atomic_fetch_add(&g_node_hist[cur], 1u, __ATOMIC_RELAXED);
            const float oo[3] = {o.x, o.y, o.z}, ii[3] = {id.x, id.y, id.z};
            float A[3], B[3];
            for (int a = 0; a < 3; ++a) {
                A[a] = bits_float(uint32_t(n.exp[a]) << 23) * ii[a];
                B[a] = (n.origin[a] - oo[a]) * ii[a];
            }
            bool hit[4];
            float entry[4], tnc[4];
            for (int k = 0; k < 4; ++k) {
                float tl[3], th[3];
                for (int a = 0; a < 3; ++a) {
                    tl[a] = fmaf((float)n.qlo[a][k], A[a], B[a]);
                    th[a] = fmaf((float)n.qhi[a][k], A[a], B[a]);
                }
                const float tn = fmaxf(fmaxf(fminf(tl[0], th[0]), fminf(tl[1], th[1])), fmaxf(fminf(tl[2], th[2]), ray.tmin));
                const float tf = fminf(fminf(fmaxf(tl[0], th[0]), fmaxf(tl[1], th[1])), fminf(fmaxf(tl[2], th[2]), best.t));
                // (the device picks entry / exit planes by the sign of the direction instead of min / max: the inverted box of an empty
                // slot never passes there, so it needs no test of its own; plane distances are finite, see DESIGN.md "Ray query semantics")
                // entry <= exit with a 1 + 2^-19 slack on the exit, as ONE fused operation: gap = entry - 1.0000019 exit <= 0 (csrc/dtraverse.h)
                tnc[k] = tn;
                const float gap = fmaf(-1.0000019f, tf, tn);
                hit[k] = n.child[k] != RPTR_BVH4_EMPTY && gap <= 0.0f;
                // order keys (csrc/dtraverse.h): a closest-hit query takes the entry distance before it is clamped to t_min (boxes the ray
                // starts inside are still told apart); an occlusion query takes `gap` -- the child the ray spends the longest stretch in comes
                // first (16.3 instead of 18.2 node visits and 3.8 instead of 5.7 triangle tests per shadow ray on the flattened forest)
                const float tn_unclamped = fmaxf(fmaxf(fminf(tl[0], th[0]), fminf(tl[1], th[1])), fminf(tl[2], th[2]));
                entry[k] = hit[k] ? (ANY ? gap : tn_unclamped) : INFINITY;
                if (g_order_key && hit[k]) { // diagnostic (tools/order_probe.py): other visit-order keys than the clamped entry distance
                    const float tn_raw = fmaxf(fmaxf(fminf(tl[0], th[0]), fminf(tl[1], th[1])), fminf(tl[2], th[2]));
                    const float tf_raw = fminf(fminf(fmaxf(tl[0], th[0]), fmaxf(tl[1], th[1])), fmaxf(tl[2], th[2]));
                    entry[k] = g_order_key == 1 ? tn_raw + tf_raw : g_order_key == 2 ? tn_raw : g_order_key == 3 ? tf_raw : g_order_key == 7 ? tn : tn + 1e-3f * tf_raw;
                }
            }
            // front to back with three comparisons of the entry distances (a miss counts as +inf, ties keep slot order): inside the pair of
            // slots (0,1), inside the pair (2,3), and the pairs against each other by their nearer member (csrc/dtraverse.h does the same
            // with conditional swaps)
            const float *e = entry;
            int order[4] = {0, 1, 2, 3};
            if (g_order_key == 5 || g_order_key == 6) { // diagnostic: ties of the entry distance (ray origin inside both boxes) go to the smaller (5) / larger (6) box
                float area[4];
                for (int k = 0; k < 4; ++k) {
                    const float dx = float(n.qhi[0][k] - n.qlo[0][k]) * bits_float(uint32_t(n.exp[0]) << 23), dy = float(n.qhi[1][k] - n.qlo[1][k]) * bits_float(uint32_t(n.exp[1]) << 23),
                                dz = float(n.qhi[2][k] - n.qlo[2][k]) * bits_float(uint32_t(n.exp[2]) << 23);
                    area[k] = (dx * dy + dy * dz + dz * dx) * (g_order_key == 5 ? 1.0f : -1.0f);
                }
                auto less = [&](int a, int b) { return e[a] < e[b] || (e[a] == e[b] && area[a] < area[b]); };
                if (less(1, 0)) std::swap(order[0], order[1]);
                if (less(3, 2)) std::swap(order[2], order[3]);
                if (less(order[2], order[0])) {
                    std::swap(order[0], order[2]);
                    std::swap(order[1], order[3]);
                }
            } else {
            if (e[1] < e[0]) std::swap(order[0], order[1]);
            if (e[3] < e[2]) std::swap(order[2], order[3]);
            if (fminf(e[2], e[3]) < fminf(e[0], e[1])) {
                std::swap(order[0], order[2]);
                std::swap(order[1], order[3]);
            }
            }
            if (g_sort_by_distance) std::stable_sort(order, order + 4, [&](int a, int b) { return e[a] < e[b]; });
            int visit[4], nv = 0;
            for (int k = 0; k < 4; ++k)
                if (hit[order[k]]) visit[nv++] = order[k];
            if (g_dead_visits && nv == 0) __atomic_fetch_add(g_dead_visits, 1ull, __ATOMIC_RELAXED);
            for (int k = nv - 1; k >= 1; --k) {
                stack_tn[sp] = tnc[visit[k]];
                stack[sp++] = n.child[visit[k]];
            }
            if (nv)
                cur = n.child[visit[0]];
            else
                pop = true;
        } else if (cur == SENTINEL) {
            cur_inst = nullptr;
            o = ray.o;
            d = ray.d;
            id = vec3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
            pop = true;
        } else {
            int first, count;
            decode_leaf(cur, first, count);
            if (cur_inst == nullptr) {
                if (count > 0) {
                    cur_inst = &bvh.insts[first];
                    if (cnt) cnt->nodes += 2; // 128-byte instance record = 2 node-sized fetches
                    o = xform_point(cur_inst->world_to_object, ray.o);
                    d = xform_dir(cur_inst->world_to_object, ray.d);
                    id = vec3(safe_rcp(d.x), safe_rcp(d.y), safe_rcp(d.z));
                    stack[sp++] = SENTINEL;
                    cur = cur_inst->blas_root;
                } else
                    pop = true;
            } else {
                for (int k = 0; k < count; ++k) {
                    const RptrBvhTri &tr = bvh.tris[first + k];
                    if (cnt) cnt->tris++;
                    float t, u, v;
                    if (!mt_intersect(o, d, vec3(tr.v0[0], tr.v0[1], tr.v0[2]), vec3(tr.e1[0], tr.e1[1], tr.e1[2]),
                                      vec3(tr.e2[0], tr.e2[1], tr.e2[2]), t, u, v))
                        continue;
                    if (!(t > ray.tmin)) continue;
                    // a world-space triangle of a flattened scene names its own instance record
                    const uint32_t rec = RPTR_BVH_TRI_INSTANCE(tr.flags);
                    const RptrBvhInstance &tri_inst = rec ? bvh.insts[rec] : *cur_inst;
                    const int ii2 = tri_inst.instance_id;
                    const bool accept = (t < best.t) || (t == best.t && best.inst >= 0 && hit_key_less(ii2, (int)tr.geom, (int)tr.prim, best));
                    if (!accept) continue;
                    if (alpha && (tr.flags & RPTR_BVH_TRI_ALPHA) && alpha->reject(tri_inst, tr, t, u, v)) continue;
                    best.t = t; best.u = u; best.v = v;
                    best.inst = ii2; best.geom = (int)tr.geom; best.prim = (int)tr.prim;
                    best.lo = o; best.ld = d;
                    if (ANY) return true;
                }
                pop = true;
            }
        }
        if (pop) {
            if (sp == 0) return best.inst >= 0;
            cur = stack[--sp];
            // diagnostic ([1], [2] of the dead-visit counters): entries that were pushed with an entry distance the hit found since lies in
            // front of -- what an entry distance kept on the stack could drop without a node step / a leaf test
            if (g_dead_visits && cur != SENTINEL && stack_tn[sp] > best.t) __atomic_fetch_add(g_dead_visits + (cur >= 0 ? 1 : 2), 1ull, __ATOMIC_RELAXED);
        }
    }
}

// brute force over the raw scene (no tree): ground truth of the canonical semantics
template <bool ANY>
static bool brute_force(const SceneView &sv, const Ray &ray, Hit &best) {
    const RptrSceneDesc *s = sv.desc;
    best.t = ray.tmax;
    best.inst = -1;
    best.u = best.v = 0; best.geom = best.prim = -1;
    for (uint32_t i = 0; i < s->num_instances; ++i) {
        const auto &in = s->instances[i];
        const auto &pm = s->parameterized_meshes[in.parameterized_mesh];
        const auto &mesh = s->meshes[pm.mesh];
        float w2o[12];
        invert_affine(in.transform, w2o);
        vec3 o = xform_point(w2o, ray.o), d = xform_dir(w2o, ray.d);
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const auto &g = s->geometries[mesh.first_geometry + j];
            for (uint32_t p = 0; p < g.num_tris; ++p) {
                vec3 v0, v1, v2;
                geom_tri(sv, g, p, v0, v1, v2);
                float t, u, v;
                if (!mt_intersect(o, d, v0, v1 - v0, v2 - v0, t, u, v)) continue;
                if (!(t > ray.tmin)) continue;
                bool accept = (t < best.t) || (t == best.t && best.inst >= 0 && hit_key_less((int)i, (int)j, (int)p, best));
                if (!accept) continue;
                best.t = t; best.u = u; best.v = v; best.inst = (int)i; best.geom = (int)j; best.prim = (int)p;
                best.lo = o; best.ld = d;
                if (ANY) return true;
            }
        }
    }
    return best.inst >= 0;
}

} // namespace orc
