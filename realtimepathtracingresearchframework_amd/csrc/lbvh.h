// lbvh.h -- device-side (re)build of a bottom-level tree and the bottom-up refit of dynamic meshes.
//
// Stands in for what the reference gets from the Vulkan driver for dynamic geometry: acceleration-structure BUILD on the GPU
// (vulkan/vulkanrt_utils.h:83-105 enqueue_build, render_vulkan.cpp:476-543; PREFER_FAST_BUILD for Mesh::Dynamic, :942-952) and UPDATE
// builds (enqueue_refit). The policy that picks between them is RenderBackendOptions::force_bvh_rebuild / rebuild_triangle_budget
// (librender/render_params.glsl.h:61,90-93) -> rptr_hip_set_bvh_policy.
//
// Build (one dynamic mesh, n triangles, everything on the caller's stream, no host round trip):
//   1. centroid bounds of the mesh (block reduction + ordered-uint atomics),
//   2. 30-bit Morton code of every centroid; key = code << 32 | triangle position  (unique keys),
//   3. radix sort of the keys (hipCUB = rocPRIM: a plain library sort),
//   4. binary radix tree over the sorted keys (Karras 2012: one thread per inner node, no atomics),
//   5. triangles gathered into sorted order,
//   6. float boxes of the binary nodes bottom-up (one thread per leaf, the second thread to arrive at a node merges its children),
//   7. 4-wide collapse: a binary node whose range holds <= RPTR_BVH_MAX_LEAF_TRIS triangles becomes a leaf; of the remaining inner
//      nodes those of even depth become 4-wide nodes that adopt their grandchildren (odd depths are absorbed) -- an exclusive scan
//      over the flags gives every 4-wide node its slot (the root gets slot 0 = the mesh's root index, as the traversal expects),
//   8. every 4-wide node is encoded with rp_bvh4_encode -- the very encoder of the host builder and the refit.
// The tree is a linear BVH: about 1.2-1.5x more node visits per ray than the host's binned-SAH tree, built in well under a
// millisecond per million triangles. Ray-query results do not depend on the tree (closest hit = smallest t, ties by ids).
//
// Refit (all dynamic meshes, one launch): a thread starts at every 4-wide node without inner children, re-encodes it from the
// triangle bounds and walks up; at a parent it counts arrivals (agent-scope acq_rel atomic: releases its own stores, acquires the
// siblings') and the last of the parent's inner children to arrive continues. Same per-node arithmetic as before (rp_refit_node),
// so "refit of unchanged vertices reproduces the built tree bit for bit" still holds.
#pragma once
#include <hipcub/hipcub.hpp>

struct RpLbvhScratch { // per scene copy, sized for the largest dynamic mesh, allocated at the first rebuild
    size_t capacity = 0; // triangles
    unsigned long long *keys_a = nullptr, *keys_b = nullptr;
    void *cub_tmp = nullptr;
    size_t cub_bytes = 0;
    int *left = nullptr, *right = nullptr, *parent = nullptr, *first = nullptr, *last = nullptr; // binary inner nodes 0..n-2
    int *leaf_parent = nullptr; // the inner node above the leaf at sorted position k
    float *bbox = nullptr;      // [n-1][6] boxes of the binary inner nodes
    uint32_t *visit = nullptr;  // arrival counters of the binary inner nodes
    uint32_t *flag = nullptr, *slot = nullptr; // 4-wide node? / its slot (exclusive scan)
    RptrBvhTri *tri_copy = nullptr;
    float *tribox_copy = nullptr;
    uint32_t *bounds = nullptr; // 6 ordered-uint encoded floats: centroid lo, hi
};

// ---- ordered-uint encoding of floats: a < b  <=>  enc(a) < enc(b)
RP_DEV uint32_t rp_ord_enc(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
RP_DEV float rp_ord_dec(uint32_t e) { return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e); }

__global__ void rp_k_lbvh_reset(uint32_t *bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;
}
// 1. centroid bounds (of 2 * centroid: lo + hi of the triangle's vertex bounds)
__global__ __launch_bounds__(256) void rp_k_lbvh_bounds(const float *tri_box, uint32_t n, uint32_t *bounds) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *b = tri_box + 6ull * i;
        for (int a = 0; a < 3; ++a) {
            const float c = b[a] + b[3 + a];
            lo[a] = fminf(lo[a], c);
            hi[a] = fmaxf(hi[a], c);
        }
    }
    for (int a = 0; a < 3; ++a) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
        }
    }
    if ((threadIdx.x & 63u) == 0) {
        for (int a = 0; a < 3; ++a) {
            if (lo[a] <= hi[a]) {
                atomicMin(&bounds[a], rp_ord_enc(lo[a]));
                atomicMax(&bounds[3 + a], rp_ord_enc(hi[a]));
            }
        }
    }
}
RP_DEV uint32_t rp_expand_bits10(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
// 2. keys
__global__ __launch_bounds__(256) void rp_k_lbvh_keys(const float *tri_box, uint32_t n, const uint32_t *bounds, unsigned long long *keys) {
    float lo[3], inv[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = rp_ord_dec(bounds[a]);
        const float ext = rp_ord_dec(bounds[3 + a]) - lo[a];
        inv[a] = ext > 0.0f ? 1023.0f / ext : 0.0f;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *b = tri_box + 6ull * i;
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            const float c = ((b[a] + b[3 + a]) - lo[a]) * inv[a];
            q[a] = (uint32_t)fminf(fmaxf(c, 0.0f), 1023.0f); // (NaN vertices land in cell 0)
        }
        const uint32_t code = (rp_expand_bits10(q[0]) << 2) | (rp_expand_bits10(q[1]) << 1) | rp_expand_bits10(q[2]);
        keys[i] = ((unsigned long long)code << 32) | (unsigned long long)i;
    }
}
// 4. binary radix tree (T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012).
// Child encoding: >= 0 binary inner node, < 0: ~k = the leaf at sorted position k.
RP_DEV int rp_lbvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j])); // keys are unique: never 64
}
__global__ __launch_bounds__(256) void rp_k_lbvh_hierarchy(const unsigned long long *keys, int n, int *left, int *right, int *parent, int *leaf_parent, int *first,
                                                           int *last) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int d = rp_lbvh_delta(keys, n, i, i + 1) - rp_lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = rp_lbvh_delta(keys, n, i, i - d);
        int lmax = 2;
        while (rp_lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
        int l = 0;
        for (int t = lmax / 2; t >= 1; t /= 2)
            if (rp_lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = rp_lbvh_delta(keys, n, i, j);
        int s = 0;
        for (int t = (l + 1) / 2; ; t = (t + 1) / 2) { // ceil(l / 2), ceil(l / 4), ... 1
            if (rp_lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
            if (t == 1) break;
        }
        const int gamma = i + s * d + min(d, 0);
        const int lo = min(i, j), hi = max(i, j);
        const int lc = lo == gamma ? ~gamma : gamma, rc = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
        left[i] = lc;
        right[i] = rc;
        first[i] = lo;
        last[i] = hi;
        if (lc >= 0) parent[lc] = i;
        else leaf_parent[gamma] = i;
        if (rc >= 0) parent[rc] = i;
        else leaf_parent[gamma + 1] = i;
        if (i == 0) parent[0] = -1;
    }
}
// 5. triangles (and their vertex bounds) into sorted order
__global__ __launch_bounds__(256) void rp_k_lbvh_gather(const unsigned long long *keys, uint32_t n, const RptrBvhTri *tri_in, const float *box_in, RptrBvhTri *tri_out,
                                                        float *box_out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t src = (uint32_t)(keys[i] & 0xFFFFFFFFull);
        const float4 *s = reinterpret_cast<const float4 *>(tri_in + src);
        float4 *d = reinterpret_cast<float4 *>(tri_out + i);
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
        const float *b = box_in + 6ull * src;
        float *o = box_out + 6ull * i;
        for (int a = 0; a < 6; ++a) o[a] = b[a];
    }
}
// 6. boxes of the binary inner nodes, bottom-up
__global__ __launch_bounds__(256) void rp_k_lbvh_boxes(const float *tri_box, int n, const int *left, const int *right, const int *parent, const int *leaf_parent,
                                                       float *bbox, uint32_t *visit) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        int p = n > 1 ? leaf_parent[k] : -1;
        while (p >= 0) {
            __threadfence();
            const uint32_t old = __hip_atomic_fetch_add(&visit[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old == 0u) break; // the first to arrive leaves the node to the second
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            const int ch[2] = {left[p], right[p]};
            for (int c = 0; c < 2; ++c) {
                const float *b = ch[c] >= 0 ? bbox + 6ull * ch[c] : tri_box + 6ull * (size_t)(~ch[c]);
                for (int a = 0; a < 3; ++a) {
                    lo[a] = fminf(lo[a], b[a]);
                    hi[a] = fmaxf(hi[a], b[3 + a]);
                }
            }
            float *o = bbox + 6ull * p;
            for (int a = 0; a < 3; ++a) {
                o[a] = lo[a];
                o[3 + a] = hi[a];
            }
            p = parent[p];
        }
    }
}
// 7a. which binary inner nodes become 4-wide nodes: inner (range > leaf size) and of even depth
__global__ __launch_bounds__(256) void rp_k_lbvh_flags(int n, const int *parent, const int *first, const int *last, uint32_t *flag) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const bool inner = last[i] - first[i] + 1 > RPTR_BVH_MAX_LEAF_TRIS;
        int depth = 0;
        for (int p = parent[i]; p >= 0; p = parent[p]) ++depth;
        flag[i] = ((inner && (depth & 1) == 0) || i == 0) ? 1u : 0u; // (the root is always a node, also of a mesh of <= 4 triangles)
    }
}
// 7b + 8. one 4-wide node per flagged binary node
RP_DEV void rp_lbvh_child(int c, const int *first, const int *last, const uint32_t *slot, int node_base, int tri_base, const float *bbox, const float *tri_box,
                          int32_t &ref, float lo[3], float hi[3], bool &inner) {
    // c: a binary child. A leaf of the binary tree, or an inner node with a small range, is a leaf of the 4-wide tree.
    int f, l;
    const float *b;
    if (c < 0) {
        f = l = ~c;
        b = tri_box + 6ull * (size_t)f;
        inner = false;
    } else {
        f = first[c];
        l = last[c];
        b = bbox + 6ull * c;
        inner = l - f + 1 > RPTR_BVH_MAX_LEAF_TRIS;
    }
    ref = inner ? node_base + (int)slot[c] : RPTR_BVH_LEAF(tri_base + f, l - f + 1);
    for (int a = 0; a < 3; ++a) {
        lo[a] = b[a];
        hi[a] = b[3 + a];
    }
}
__global__ __launch_bounds__(256) void rp_k_lbvh_emit(int n, const int *left, const int *right, const int *first, const int *last, const uint32_t *flag,
                                                      const uint32_t *slot, const float *bbox, const float *tri_box, int node_base, int tri_base,
                                                      RptrBvh4Node *nodes, float *node_box, int *parent4, uint32_t *ninner4, uint32_t *visit4, int *out_count) {
    const int n_inner = max(n - 1, 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inner; i += gridDim.x * blockDim.x) {
        if (n < 2) { // a mesh of one triangle (or none): a root with that leaf
            int32_t child[4] = {n == 1 ? RPTR_BVH_LEAF(tri_base, 1) : RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY};
            RpBox4 b;
            for (int k = 0; k < 4; ++k)
                for (int a = 0; a < 3; ++a) {
                    b.lo[k][a] = (k == 0 && n == 1) ? tri_box[a] : INFINITY;
                    b.hi[k][a] = (k == 0 && n == 1) ? tri_box[3 + a] : -INFINITY;
                }
            RptrBvh4Node nd;
            float *nb = node_box + 6ull * node_base;
            rp_bvh4_encode(b, child, &nd, nb, nb + 3);
            nodes[node_base] = nd;
            parent4[node_base] = -1;
            ninner4[node_base] = 0;
            visit4[node_base] = 0;
            *out_count = 1;
            return;
        }
        if (i == n - 2) *out_count = (int)(slot[i] + flag[i]); // (the scan is exclusive: the last element closes the count)
        if (!flag[i]) continue;
        const int me = node_base + (int)slot[i];
        int32_t child[4] = {RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY};
        RpBox4 b;
        for (int k = 0; k < 4; ++k)
            for (int a = 0; a < 3; ++a) {
                b.lo[k][a] = INFINITY;
                b.hi[k][a] = -INFINITY;
            }
        int nc = 0;
        uint32_t n_in = 0;
        const bool me_inner = last[i] - first[i] + 1 > RPTR_BVH_MAX_LEAF_TRIS;
        if (!me_inner) { // only the root of a tiny mesh: one leaf with everything
            child[0] = RPTR_BVH_LEAF(tri_base + first[i], last[i] - first[i] + 1);
            for (int a = 0; a < 3; ++a) {
                b.lo[0][a] = bbox[6ull * i + a];
                b.hi[0][a] = bbox[6ull * i + 3 + a];
            }
            nc = 1;
        } else {
            const int two[2] = {left[i], right[i]};
            for (int c = 0; c < 2; ++c) {
                const int ch = two[c];
                const bool absorb = ch >= 0 && last[ch] - first[ch] + 1 > RPTR_BVH_MAX_LEAF_TRIS; // an inner node of odd depth: its children move up
                const int cand[2] = {absorb ? left[ch] : ch, absorb ? right[ch] : ch};
                for (int g = 0; g < (absorb ? 2 : 1); ++g) {
                    bool inner;
                    rp_lbvh_child(cand[g], first, last, slot, node_base, tri_base, bbox, tri_box, child[nc], b.lo[nc], b.hi[nc], inner);
                    if (inner) {
                        parent4[child[nc]] = me;
                        ++n_in;
                    }
                    ++nc;
                }
            }
        }
        RptrBvh4Node nd;
        float *nb = node_box + 6ull * me;
        rp_bvh4_encode(b, child, &nd, nb, nb + 3);
        nodes[me] = nd;
        ninner4[me] = n_in;
        visit4[me] = 0;
        if (i == 0) parent4[me] = -1;
    }
}

// ------------------------------------------------------------------ bottom-up refit of the dynamic bottom-level trees
// meshes[]: (node_base, pointer to the node count) of every dynamic mesh that is refitted by this launch
struct RpRefitMesh {
    int node_base;
    int node_count; // host-known count, or -1: read *count_ptr (a tree the device built)
    const int *count_ptr;
};
__global__ __launch_bounds__(256) void rp_k_refit_up(RptrBvh4Node *nodes, float *node_box, const float *tri_box, const int *parent4, const uint32_t *ninner4,
                                                     uint32_t *visit4, RpRefitMesh mesh) {
    const int count = mesh.node_count >= 0 ? mesh.node_count : *mesh.count_ptr;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        int j = mesh.node_base + k;
        if (ninner4[j] != 0u) continue; // starts are the nodes whose children are all leaves
        for (;;) {
            rp_refit_node(nodes, node_box, tri_box, nullptr, (uint32_t)j);
            const int p = parent4[j];
            if (p < 0) break;
            __threadfence();
            const uint32_t old = __hip_atomic_fetch_add(&visit4[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1u) % ninner4[p] != 0u) break; // not the last inner child to arrive (the counter is never reset: it runs modulo)
            j = p;
        }
    }
}
