// ploc.h -- device-side BUILD of the acceleration structure of static geometry (set_scene), SAH-like quality at sort speed.
//
// Stands in for what the reference gets from the Vulkan driver inside set_scene: BLAS builds on the GPU with PREFER_FAST_TRACE for
// static meshes + compaction (vulkan/vulkanrt_utils.h:83-105 enqueue_build / compact, vulkan/render_vulkan.cpp:476-543,942-952). The
// host builder (bvh_build.cpp: binned SAH) needs 5.5 s for the 10 M world-space triangles of a flattened forest; the linear BVH of
// lbvh.h is built in milliseconds but costs 38 % more node visits per ray there. This builder gets within 5 % of the host tree:
//
//   1. world-space (flattened scene) or object-space (one mesh) triangles + vertex bounds, straight from the quantised vertex streams
//      that shading reads (rp_k_build_tris: the host loop of build_host_bvh, one thread per triangle, same float operations),
//   2. Morton keys on a cubic grid, radix sort, gather (lbvh.h steps 1-3, 5),
//   3. PLOC (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG 2018): the
//      clusters -- first the triangles in Morton order -- each look RP_PLOC_RADIUS positions to both sides for the neighbour whose
//      union with them has the smallest surface area; mutual nearest neighbours merge into a new node (the lower position keeps the
//      slot, an exclusive scan compacts the rest and numbers the new nodes: no atomics, the tree does not depend on scheduling);
//      repeated until RP_PLOC_TOP clusters are left,
//   4. the TOP of the tree -- where local clustering is weakest (+9 % node visits on the forest, +20 % on the height field when PLOC
//      runs to the root) -- is a binned-SAH tree over the remaining <= 65 536 cluster boxes, built by the host builder in
//      milliseconds and stitched on,
//   5. depth-first positions of all triangles (every subtree a contiguous range: what the leaf encoding needs), second gather,
//   6. the 4-wide collapse by the host builder's rule -- the AREA-OPTIMAL collapse (bvh_build.h COLLAPSE_OPTIMAL: dynamic programming over
//      "subtree shown through at most i slots"; the costs come bottom-up with the clustering iterations, rp_ploc_dp_node / rp_ploc_expand) --
//      breadth first: one launch pair per depth level, node indices from exclusive scans; then boxes + encoding level by level, deepest
//      first, with the refit code of every other tree.
// bvh_build.cpp: build_bvh2_ploc is the same algorithm stated on the host (same keys, same float operations, same tie rules): both
// give the same tree, which is how the device path is tested (tests/test_gpu_device_build.py) and how its quality was measured before
// it was written (profiles/r03_notes.md: forest 27.2 node visits per closest-hit ray against 26.0 for the host SAH tree and 35.8 for
// the linear BVH; height field 9.12 / 8.91 / 9.5).
#pragma once
#include "bvh_build.h"
#include "lbvh.h"

#ifndef RP_PLOC_RADIUS
#define RP_PLOC_RADIUS 25
#endif
#ifndef RP_PLOC_TOP
#define RP_PLOC_TOP RP_PLOC_TOP_DEFAULT // (bvh_build.h)
#endif

// one source of triangles of a build: `count` triangles of one geometry (optionally under an instance transform), output positions
// [begin, begin + count)
struct RpBuildSegment {
    const uint64_t *qpos;
    const uint8_t *mat_ids;  // per-triangle material ids of the geometry under this parameterized mesh, or NULL
    float scaling[3];
    int32_t material_offset;
    float offset[3];
    uint32_t begin;
    float transform[12];     // row-major 3x4 (world-space builds), unused otherwise
    uint32_t count, geom, flags_hi /* (instance record index) << 8, or 0 */, has_transform;
};

// 1. triangles of a build from the vertex streams (what build_host_bvh does on the host: dequantise, transform, edges, bounds)
__global__ __launch_bounds__(256) void rp_k_build_tris(const RpBuildSegment *segs, int n_segs, uint32_t n, const uint8_t *mat_alpha, uint32_t n_mats, RptrBvhTri *tris,
                                                       float *tri_box) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int lo = 0, hi = n_segs - 1; // the segment that holds output triangle i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (segs[mid].begin <= i) lo = mid;
            else hi = mid - 1;
        }
        const RpBuildSegment &sg = segs[lo];
        const uint32_t t = i - sg.begin;
        float w[3][3];
        for (int k = 0; k < 3; ++k) {
            const uint64_t q = sg.qpos[3ull * t + k];
            const float v[3] = {float(uint32_t(q) & 0x1FFFFFu) * sg.scaling[0] + sg.offset[0], float(uint32_t(q >> 21) & 0x1FFFFFu) * sg.scaling[1] + sg.offset[1],
                                float(uint32_t(q >> 42) & 0x1FFFFFu) * sg.scaling[2] + sg.offset[2]};
            if (sg.has_transform) {
                const float *M = sg.transform;
                for (int r = 0; r < 3; ++r) w[k][r] = ((M[4 * r] * v[0] + M[4 * r + 1] * v[1]) + M[4 * r + 2] * v[2]) + M[4 * r + 3];
            } else
                for (int r = 0; r < 3; ++r) w[k][r] = v[r];
        }
        RptrBvhTri tri;
        float *b = tri_box + 6ull * i;
        for (int k = 0; k < 3; ++k) {
            tri.v0[k] = w[0][k];
            tri.e1[k] = w[1][k] - w[0][k];
            tri.e2[k] = w[2][k] - w[0][k];
            b[k] = fminf(w[0][k], fminf(w[1][k], w[2][k]));
            b[3 + k] = fmaxf(w[0][k], fmaxf(w[1][k], w[2][k]));
        }
        const int64_t mid = (int64_t)sg.material_offset + (sg.mat_ids ? (int64_t)sg.mat_ids[t] : 0);
        const bool alpha = mid >= 0 && mid < (int64_t)n_mats && mat_alpha[mid] != 0;
        tri.prim = t;
        tri.geom = sg.geom;
        tri.flags = (alpha ? RPTR_BVH_TRI_ALPHA : 0u) | sg.flags_hi;
        float4 *d = reinterpret_cast<float4 *>(tris + i);
        const float4 *s = reinterpret_cast<const float4 *>(&tri);
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
    }
}

// ---- 3. PLOC. Node ids: 0..n-1 the triangles in Morton order, n.. the merges in creation order. left / right are indexed by id - n.
__global__ __launch_bounds__(256) void rp_k_ploc_init(uint32_t n, const float *tri_box, uint32_t *cid, float *cbox, int *parent, uint32_t *count) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        cid[i] = i;
        for (int a = 0; a < 6; ++a) cbox[6ull * i + a] = tri_box[6ull * i + a];
        parent[i] = -1;
        count[i] = 1u;
    }
}
// nearest neighbour of every cluster within RADIUS positions: smallest half area of the union, ties to the lower position
template <int RADIUS>
__global__ __launch_bounds__(256) void rp_k_ploc_nn(uint32_t m, const float *cbox, uint32_t *nn) {
    __shared__ float s_box[6][256 + 2 * RADIUS];
    const int64_t base = (int64_t)blockIdx.x * 256 - RADIUS;
    for (int k = threadIdx.x; k < 256 + 2 * RADIUS; k += 256) {
        const int64_t j = base + k;
        const bool in = j >= 0 && j < (int64_t)m;
        for (int a = 0; a < 6; ++a) s_box[a][k] = in ? cbox[6ull * (uint64_t)j + a] : 0.0f;
    }
    __syncthreads();
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= m) return;
    const int me = threadIdx.x + RADIUS;
    const float lo0 = s_box[0][me], lo1 = s_box[1][me], lo2 = s_box[2][me], hi0 = s_box[3][me], hi1 = s_box[4][me], hi2 = s_box[5][me];
    float best = INFINITY;
    uint32_t arg = i == 0 ? min(m - 1, (uint32_t)RADIUS) : (i > (uint32_t)RADIUS ? i - RADIUS : 0u); // (no candidate beats +inf: NaN boxes)
    for (int k = -RADIUS; k <= RADIUS; ++k) {
        if (k == 0) continue;
        const int64_t j = (int64_t)i + k;
        if (j < 0 || j >= (int64_t)m) continue;
        const int at = me + k;
        const float dx = fmaxf(hi0, s_box[3][at]) - fminf(lo0, s_box[0][at]), dy = fmaxf(hi1, s_box[4][at]) - fminf(lo1, s_box[1][at]),
                    dz = fmaxf(hi2, s_box[5][at]) - fminf(lo2, s_box[2][at]);
        const float a = dx * dy + dy * dz + dz * dx;
        if (a < best) {
            best = a;
            arg = (uint32_t)j;
        }
    }
    nn[i] = arg;
}
// keep (low word): the cluster stays in the list (it is not the upper partner of a merge); merge (high word): it is the lower partner
__global__ __launch_bounds__(256) void rp_k_ploc_flags(uint32_t m, const uint32_t *nn, unsigned long long *packed) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t j = nn[i];
        const bool mutual = nn[j] == i;
        packed[i] = (unsigned long long)((mutual && i > j) ? 0u : 1u) | ((unsigned long long)((mutual && i < j) ? 1u : 0u) << 32);
    }
}
// totals[0] = clusters after this iteration, totals[1] = nodes made so far (in / out)
__global__ __launch_bounds__(256) void rp_k_ploc_apply(uint32_t m, uint32_t n, const uint32_t *nn, const unsigned long long *packed, const unsigned long long *scan,
                                                       const uint32_t *cid_in, const float *cbox_in, uint32_t *cid_out, float *cbox_out, int *left, int *right,
                                                       int *parent, uint32_t *count, float *area, const uint32_t *totals_in, uint32_t *totals_out) {
    const uint32_t made = totals_in[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const unsigned long long p = packed[i], s = scan[i];
        if (i == m - 1) {
            totals_out[0] = (uint32_t)(s & 0xFFFFFFFFull) + (uint32_t)(p & 0xFFFFFFFFull);
            totals_out[1] = made + (uint32_t)(s >> 32) + (uint32_t)(p >> 32);
        }
        if ((p & 0xFFFFFFFFull) == 0) continue;
        const uint32_t slot = (uint32_t)(s & 0xFFFFFFFFull);
        float b[6];
        for (int a = 0; a < 6; ++a) b[a] = cbox_in[6ull * i + a];
        uint32_t id = cid_in[i];
        if (p >> 32) {
            const uint32_t j = nn[i], a_id = id, b_id = cid_in[j];
            for (int a = 0; a < 3; ++a) {
                b[a] = fminf(b[a], cbox_in[6ull * j + a]);
                b[3 + a] = fmaxf(b[3 + a], cbox_in[6ull * j + 3 + a]);
            }
            id = made + (uint32_t)(s >> 32);
            left[id - n] = (int)a_id;
            right[id - n] = (int)b_id;
            parent[a_id] = (int)id;
            parent[b_id] = (int)id;
            parent[id] = -1;
            count[id] = count[a_id] + count[b_id];
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            area[id - n] = (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? dx * dy + dy * dz + dz * dx : 0.0f; // (what the collapse weighs slots by)
        }
        cid_out[slot] = id;
        for (int a = 0; a < 6; ++a) cbox_out[6ull * slot + a] = b[a];
    }
}
__global__ __launch_bounds__(256) void rp_k_ploc_gather_counts(uint32_t m, const uint32_t *cid, const uint32_t *count, uint32_t *out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = count[cid[i]];
}
// 4. the stitched-on top: nodes first_id.. (left, right, count given per node), parents of everything they refer to
__global__ __launch_bounds__(256) void rp_k_ploc_stitch(uint32_t k, uint32_t n, uint32_t first_id, const int *top_left, const int *top_right, const uint32_t *top_count,
                                                        const float *top_area, int *left, int *right, int *parent, uint32_t *count, float *area) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < k; i += gridDim.x * blockDim.x) {
        const uint32_t id = first_id + i;
        left[id - n] = top_left[i];
        right[id - n] = top_right[i];
        count[id] = top_count[i];
        area[id - n] = top_area[i];
        parent[top_left[i]] = (int)id;
        parent[top_right[i]] = (int)id;
        if (i == k - 1) parent[id] = -1; // the root is made last
    }
}
// 5. depth-first position of the first triangle below a node: the triangles of the left siblings on the way up
RP_DEV uint32_t rp_ploc_first(uint32_t id, uint32_t n, const int *left, const int *right, const int *parent, const uint32_t *count) {
    uint32_t off = 0, c = id;
    for (int p = parent[c]; p >= 0; p = parent[p]) {
        if ((uint32_t)right[(uint32_t)p - n] == c) off += count[left[(uint32_t)p - n]];
        c = (uint32_t)p;
    }
    return off;
}
// depth-first position of the first triangle below every inner node (index id - n)
__global__ __launch_bounds__(256) void rp_k_ploc_firsts(uint32_t n, const int *left, const int *right, const int *parent, const uint32_t *count, uint32_t *nfirst) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k + 1 < n; k += gridDim.x * blockDim.x) nfirst[k] = rp_ploc_first(n + k, n, left, right, parent, count);
}

// ---- 6. the 4-wide collapse, breadth first (the host builder's rule, bvh_build.cpp collapse_bvh4 COLLAPSE_OPTIMAL, below). A subtree of
// <= RP_LBVH_LEAF_TRIS triangles is a leaf. One launch pair per 4-wide depth level: levels are what the refit needs anyway, and a node's index -- its level's
// base + its position in the level's queue, given by an exclusive scan -- does not depend on scheduling.
RP_DEV bool rp_ploc_expandable(int id, uint32_t n, const uint32_t *count) { return (uint32_t)id >= n && count[id] > (uint32_t)RP_LBVH_LEAF_TRIS; }
// The collapse is the host builder's (bvh_build.h COLLAPSE_OPTIMAL): the one that minimises the summed surface area of the binary nodes that
// survive as wide nodes. dpc[id - n] = (cost of subtree id as ONE wide node, cost of its children spread over at most 2 / 3 / 4 slots of a
// parent), computed bottom-up: rp_k_ploc_dp_range once per clustering iteration (a merge's children are older than the merge), the stitched
// top on the host (rptr_hip.hip). Same float operations in the same order as bvh_build.cpp collapse_bvh4: both give the same tree.
RP_DEV float rp_ploc_dp_g(int x, int i, uint32_t n, const uint32_t *count, const float4 *dpc) { // subtree x through at most i slots
    if (!rp_ploc_expandable(x, n, count)) return 0.0f;
    const float4 c = dpc[(uint32_t)x - n];
    return i == 1 ? c.x : fminf(c.x, i == 2 ? c.y : i == 3 ? c.z : c.w);
}
RP_DEV float4 rp_ploc_dp_node(int l, int r, float a, uint32_t n, const uint32_t *count, const float4 *dpc) {
    const float l1 = rp_ploc_dp_g(l, 1, n, count, dpc), l2 = rp_ploc_dp_g(l, 2, n, count, dpc), l3 = rp_ploc_dp_g(l, 3, n, count, dpc);
    const float r1 = rp_ploc_dp_g(r, 1, n, count, dpc), r2 = rp_ploc_dp_g(r, 2, n, count, dpc), r3 = rp_ploc_dp_g(r, 3, n, count, dpc);
    const float f2 = l1 + r1, f3 = fminf(l1 + r2, l2 + r1), f4 = fminf(fminf(l1 + r3, l2 + r2), l3 + r1);
    return make_float4(a + f4, f2, f3, f4);
}
__global__ __launch_bounds__(256) void rp_k_ploc_dp_range(uint32_t begin, uint32_t end, uint32_t n, const int *left, const int *right, const uint32_t *count,
                                                          const float *area, float4 *dpc) {
    for (uint32_t id = begin + blockIdx.x * blockDim.x + threadIdx.x; id < end; id += gridDim.x * blockDim.x)
        if (rp_ploc_expandable((int)id, n, count)) dpc[id - n] = rp_ploc_dp_node(left[id - n], right[id - n], area[id - n], n, count, dpc);
}
__global__ __launch_bounds__(256) void rp_k_ploc_gather_costs(uint32_t m, uint32_t n, const uint32_t *cid, const float4 *dpc, float4 *out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) out[i] = cid[i] >= n ? dpc[cid[i] - n] : make_float4(0.f, 0.f, 0.f, 0.f);
}
// the children of wide node b: its binary node is dissolved over four slots; a subtree below keeps its root (one slot) or is dissolved in turn,
// whichever the costs say (ties: keep the root; the split of a quota: the first of the cheapest). Slots come out left to right.
RP_DEV int rp_ploc_expand(int b, uint32_t n, const int *left, const int *right, const uint32_t *count, const float4 *dpc, int slots[4]) {
    int ns = 0, sp = 0;
    int wn[4], wq[4];
    auto split = [&](int c, int q) {
        const int l = left[(uint32_t)c - n], r = right[(uint32_t)c - n];
        int best_l = 1;
        float best = INFINITY;
        for (int k = 1; k < q; ++k) {
            const float v = rp_ploc_dp_g(l, k, n, count, dpc) + rp_ploc_dp_g(r, q - k, n, count, dpc);
            if (v < best) {
                best = v;
                best_l = k;
            }
        }
        wn[sp] = r;
        wq[sp++] = q - best_l;
        wn[sp] = l;
        wq[sp++] = best_l;
    };
    split(b, 4);
    while (sp > 0) {
        const int c = wn[--sp], q = wq[sp];
        bool keep = !rp_ploc_expandable(c, n, count) || q == 1;
        if (!keep) {
            const float4 d = dpc[(uint32_t)c - n];
            keep = d.x <= (q == 2 ? d.y : q == 3 ? d.z : d.w);
        }
        if (keep)
            slots[ns++] = c;
        else
            split(c, q);
    }
    return ns;
}
// pass 1 of a level: how many inner children each of its nodes has
__global__ __launch_bounds__(256) void rp_k_ploc_collapse_count(const int *queue, uint32_t size, uint32_t n, const int *left, const int *right, const uint32_t *count,
                                                                const float4 *dpc, uint32_t *inner) {
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < size; w += gridDim.x * blockDim.x) {
        int slots[4];
        const int ns = rp_ploc_expand(queue[w], n, left, right, count, dpc, slots);
        uint32_t c = 0;
        for (int k = 0; k < ns; ++k) c += rp_ploc_expandable(slots[k], n, count) ? 1u : 0u;
        inner[w] = c;
    }
}
// pass 2: child references (inner: next level's base + scan position; leaf: its triangle range in depth-first order) and the next queue.
// The root of a tree whose triangles all fit one leaf is a node with that one leaf (queue[0] = the root, flagged by `tiny`).
__global__ __launch_bounds__(256) void rp_k_ploc_collapse_emit(const int *queue, uint32_t size, uint32_t n, const int *left, const int *right, const int *parent,
                                                               const uint32_t *count, const float4 *dpc, const uint32_t *nfirst, const uint32_t *inner_scan,
                                                               uint32_t level_base, uint32_t next_base, RptrBvh4Node *nodes, int *next_queue, uint32_t *next_size) {
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < size; w += gridDim.x * blockDim.x) {
        int slots[4];
        const int b = queue[w];
        int32_t child[4] = {RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY};
        uint32_t at = inner_scan[w];
        if (!rp_ploc_expandable(b, n, count)) // (only the root of a tiny tree gets here)
            child[0] = RPTR_BVH_LEAF(0, (int)count[b]);
        else {
            const int ns = rp_ploc_expand(b, n, left, right, count, dpc, slots);
            for (int k = 0; k < ns; ++k) {
                const int x = slots[k];
                if (rp_ploc_expandable(x, n, count)) {
                    next_queue[at] = x;
                    child[k] = (int32_t)(next_base + at);
                    ++at;
                } else {
                    const uint32_t first = (uint32_t)x < n ? rp_ploc_first((uint32_t)x, n, left, right, parent, count) : nfirst[(uint32_t)x - n];
                    child[k] = RPTR_BVH_LEAF((int)first, (int)count[x]);
                }
            }
        }
        if (w == size - 1) *next_size = at;
        RptrBvh4Node nd;
        __builtin_memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 4; ++k) nd.child[k] = child[k];
        nodes[level_base + w] = nd;
    }
}
// boxes + encoding of one level (children of deeper levels are done): the refit of every other tree (kernels_misc.h rp_refit_node)
__global__ __launch_bounds__(256) void rp_k_ploc_refit_range(RptrBvh4Node *nodes, float *node_box, const float *tri_box, uint32_t begin, uint32_t end) {
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x) rp_refit_node(nodes, node_box, tri_box, nullptr, i);
}
__global__ __launch_bounds__(256) void rp_k_ploc_scatter(uint32_t n, const int *left, const int *right, const int *parent, const uint32_t *count, const RptrBvhTri *tri_in,
                                                         const float *box_in, RptrBvhTri *tri_out, float *box_out) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        const uint32_t pos = rp_ploc_first(k, n, left, right, parent, count);
        const float4 *s = reinterpret_cast<const float4 *>(tri_in + k);
        float4 *d = reinterpret_cast<float4 *>(tri_out + pos);
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
        for (int a = 0; a < 6; ++a) box_out[6ull * pos + a] = box_in[6ull * k + a];
    }
}
