// lbvh.h -- device-side (re)build of a bottom-level tree and the bottom-up refit of dynamic meshes.
//
// Stands in for what the reference gets from the Vulkan driver for dynamic geometry: acceleration-structure BUILD on the GPU
// (vulkan/vulkanrt_utils.h:83-105 enqueue_build, render_vulkan.cpp:476-543; PREFER_FAST_BUILD for Mesh::Dynamic, :942-952) and UPDATE
// builds (enqueue_refit). The policy that picks between them is RenderBackendOptions::force_bvh_rebuild / rebuild_triangle_budget
// (librender/render_params.glsl.h:61,90-93) -> rptr_hip_set_bvh_policy.
//
// Build (one dynamic mesh, n triangles, everything on the caller's stream, no host round trip):
//   1. centroid bounds of the mesh (block reduction + ordered-uint atomics),
//   2. Morton code of every centroid (as many bits per axis as the key has room for); key = code << index bits | triangle position,
//   3. radix sort of the keys (hipCUB = rocPRIM: a plain library sort),
//   4. binary radix tree over the sorted keys (Karras 2012: one thread per inner node, no atomics),
//   5. triangles gathered into sorted order,
//   6. 4-wide collapse: a binary node whose range holds <= RP_LBVH_LEAF_TRIS triangles becomes a leaf; of the remaining inner nodes
//      those of even depth become 4-wide nodes that adopt their grandchildren (odd depths are absorbed) -- an exclusive scan over the
//      flags gives every 4-wide node its slot (the root gets slot 0 = the mesh's root index, as the traversal expects),
//   7. the child references of every 4-wide node are written, and the nodes are listed by depth (every inner child of a 4-wide node
//      lies exactly one 4-wide level below it: depth levels are a valid bottom-up order),
//   8. boxes and encoding come from a REFIT of the new topology, deepest level first: rp_refit_node + rp_bvh4_encode -- the very
//      code of every other refit and the encoder of the host builder.
// No step synchronises threads through memory (per-XCD L2s are not coherent: agent-scope fences inside a kernel cost a write-back
// each -- a bottom-up pass with arrival counters took 0.8 ms per refit of 300 k nodes against 0.1 ms for launches per level).
// The tree is a linear BVH over cubic Morton cells: on the 1 M-triangle height field as good as the host's binned-SAH tree (9.5 against
// 9.4 node visits per ray, profiles/r02_notes.md), in general looser; built in about three milliseconds per million triangles. Ray-query results do not depend on the tree (closest hit = smallest t, ties by ids).
//
// Refit: launches per depth level, deepest first; the level bounds are read from the device (a device-built tree's level sizes are
// unknown to the host until an asynchronous copy has arrived; until then every possible level gets its launch).
#pragma once
#include <hipcub/hipcub.hpp>

#define RP_REFIT_LEVELS 40 // 4-wide depth levels a tree can have (64-bit keys: binary depth <= 64, 4-wide depth <= 32)
#ifndef RP_LBVH_LEAF_TRIS
#define RP_LBVH_LEAF_TRIS 2 // triangles per leaf of a device-built tree (Morton-order leaves are looser than SAH leaves: fewer per leaf)
#endif

struct RpLbvhScratch { // per scene copy, sized for the largest dynamic mesh, allocated at the first rebuild
    size_t capacity = 0; // triangles
    unsigned long long *keys_a = nullptr, *keys_b = nullptr;
    void *cub_tmp = nullptr;
    size_t cub_bytes = 0;
    int *left = nullptr, *right = nullptr, *parent = nullptr, *first = nullptr, *last = nullptr; // binary inner nodes 0..n-2
    uint32_t *flag = nullptr, *slot = nullptr, *depth4 = nullptr; // 4-wide node? / its slot (exclusive scan) / its 4-wide depth
    uint32_t *level_hist = nullptr, *level_cursor = nullptr;      // RP_REFIT_LEVELS entries each
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
RP_DEV unsigned long long rp_expand_bits21(uint32_t v) { // 21 bits -> every third bit of 63
    unsigned long long x = v & 0x1FFFFFull;
    x = (x | (x << 32)) & 0x1F00000000FFFFull;
    x = (x | (x << 16)) & 0x1F0000FF0000FFull;
    x = (x | (x << 8)) & 0x100F00F00F00F00Full;
    x = (x | (x << 4)) & 0x10C30C30C30C30C3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
// 2. keys
// key = Morton code of the centroid in the high bits | triangle position in the low `index_bits` bits (unique keys). The code gets
// all the bits the index leaves: 14 per axis for a million triangles
__global__ __launch_bounds__(256) void rp_k_lbvh_keys(const float *tri_box, uint32_t n, const uint32_t *bounds, unsigned long long *keys, int index_bits) {
    const int axis_bits = min(21, (64 - index_bits) / 3);
    const float cells = (float)((1u << axis_bits) - 1u);
    // CUBIC cells: one scale for the three axes. Scaling every axis to the full range makes the curve split a flat mesh along its thin
    // axis at every third level (a height field by height: children that overlap completely in plan).
    float lo[3], max_ext = 0.0f;
    for (int a = 0; a < 3; ++a) {
        lo[a] = rp_ord_dec(bounds[a]);
        max_ext = fmaxf(max_ext, rp_ord_dec(bounds[3 + a]) - lo[a]);
    }
    const float inv = max_ext > 0.0f ? cells / max_ext : 0.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float *b = tri_box + 6ull * i;
        uint32_t q[3];
        for (int a = 0; a < 3; ++a) {
            const float c = ((b[a] + b[3 + a]) - lo[a]) * inv;
            q[a] = (uint32_t)fminf(fmaxf(c, 0.0f), cells); // (NaN vertices land in cell 0)
        }
        // 21 bits per axis interleaved, then cut down to the 3 * axis_bits that are there (the low bits of the 21 are zero-filled cells)
        const unsigned long long code = (rp_expand_bits21(q[0]) << 2) | (rp_expand_bits21(q[1]) << 1) | rp_expand_bits21(q[2]);
        keys[i] = (code << index_bits) | (unsigned long long)i;
    }
}
// 4. binary radix tree (T. Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", HPG 2012).
// Child encoding: >= 0 binary inner node, < 0: ~k = the leaf at sorted position k.
RP_DEV int rp_lbvh_delta(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j])); // keys are unique: never 64
}
__global__ __launch_bounds__(256) void rp_k_lbvh_hierarchy(const unsigned long long *keys, int n, int *left, int *right, int *parent, int *first, int *last) {
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
        if (rc >= 0) parent[rc] = i;
        if (i == 0) parent[0] = -1;
    }
}
// 5. triangles (and their vertex bounds) into sorted order
__global__ __launch_bounds__(256) void rp_k_lbvh_gather(const unsigned long long *keys, uint32_t n, const RptrBvhTri *tri_in, const float *box_in, RptrBvhTri *tri_out,
                                                        float *box_out, unsigned long long index_mask) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t src = (uint32_t)(keys[i] & index_mask);
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
// 6. which binary inner nodes become 4-wide nodes: inner (range > leaf size) and of even depth. depth4[i] = depth / 2 for those.
__global__ __launch_bounds__(256) void rp_k_lbvh_flags(int n, const int *parent, const int *first, const int *last, uint32_t *flag, uint32_t *depth4) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const bool inner = last[i] - first[i] + 1 > RP_LBVH_LEAF_TRIS;
        int depth = 0;
        for (int p = parent[i]; p >= 0; p = parent[p]) ++depth;
        const bool node = (inner && (depth & 1) == 0) || i == 0; // (the root is always a node, also of a mesh of <= RP_LBVH_LEAF_TRIS triangles)
        flag[i] = node ? 1u : 0u;
        depth4[i] = (uint32_t)(depth >> 1);
    }
}
// 7. topology of one 4-wide node per flagged binary node: child references only -- boxes and encoding come from the refit that
// follows (rp_refit_node: leaf children from the triangle bounds, inner children from the exact float bounds of the level below).
// A binary child that is a leaf of the binary tree, or an inner node with a small range, is a leaf of the 4-wide tree.
RP_DEV int32_t rp_lbvh_child_ref(int c, const int *first, const int *last, const uint32_t *slot, int node_base, int tri_base) {
    if (c < 0) return RPTR_BVH_LEAF(tri_base + ~c, 1);
    const int f = first[c], l = last[c];
    return l - f + 1 > RP_LBVH_LEAF_TRIS ? node_base + (int)slot[c] : RPTR_BVH_LEAF(tri_base + f, l - f + 1);
}
// levels: slot k of the level table holds the nodes of depth RP_REFIT_LEVELS - 1 - k, so that ascending k = deepest first
__global__ __launch_bounds__(256) void rp_k_lbvh_emit(int n, const int *left, const int *right, const int *first, const int *last, const uint32_t *flag,
                                                      const uint32_t *slot, const uint32_t *depth4, int node_base, int tri_base, RptrBvh4Node *nodes,
                                                      uint32_t *level_hist, int *out_count) {
    __shared__ uint32_t s_hist[RP_REFIT_LEVELS];
    if (threadIdx.x < RP_REFIT_LEVELS) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const int n_inner = max(n - 1, 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inner; i += gridDim.x * blockDim.x) {
        int32_t child[4] = {RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY, RPTR_BVH4_EMPTY};
        int me = node_base;
        uint32_t depth = 0;
        if (n < 2) { // a mesh of one triangle (or none): a root with that leaf
            if (n == 1) child[0] = RPTR_BVH_LEAF(tri_base, 1);
            *out_count = 1;
        } else {
            if (i == n - 2) *out_count = (int)(slot[i] + flag[i]); // (the scan is exclusive: the last element closes the count)
            if (!flag[i]) continue;
            me = node_base + (int)slot[i];
            depth = depth4[i];
            if (last[i] - first[i] + 1 <= RP_LBVH_LEAF_TRIS) // only the root of a tiny mesh: one leaf with everything
                child[0] = RPTR_BVH_LEAF(tri_base + first[i], last[i] - first[i] + 1);
            else {
                int nc = 0;
                const int two[2] = {left[i], right[i]};
                for (int c = 0; c < 2; ++c) {
                    const int ch = two[c];
                    if (ch >= 0 && last[ch] - first[ch] + 1 > RP_LBVH_LEAF_TRIS) { // an inner node of odd depth: its children move up
                        child[nc++] = rp_lbvh_child_ref(left[ch], first, last, slot, node_base, tri_base);
                        child[nc++] = rp_lbvh_child_ref(right[ch], first, last, slot, node_base, tri_base);
                    } else
                        child[nc++] = rp_lbvh_child_ref(ch, first, last, slot, node_base, tri_base);
                }
            }
        }
        RptrBvh4Node nd;
        __builtin_memset(&nd, 0, sizeof(nd));
        for (int k = 0; k < 4; ++k) nd.child[k] = child[k];
        nd._pad1[0] = min(depth, (uint32_t)(RP_REFIT_LEVELS - 1)); // parked in the node's padding until rp_k_lbvh_level_scatter has read it
        nodes[me] = nd;
        atomicAdd(&s_hist[RP_REFIT_LEVELS - 1 - min(depth, (uint32_t)(RP_REFIT_LEVELS - 1))], 1u);
    }
    __syncthreads();
    if (threadIdx.x < RP_REFIT_LEVELS && s_hist[threadIdx.x]) atomicAdd(&level_hist[threadIdx.x], s_hist[threadIdx.x]);
}
// level table of one mesh from the histogram: [begin, end) into the mesh's slice of the node list; cursor[k] = begin (for the scatter)
__global__ void rp_k_lbvh_level_scan(const uint32_t *level_hist, uint32_t list_base, uint2 *levels, uint32_t *cursor) {
    if (threadIdx.x != 0) return;
    uint32_t at = list_base;
    for (int k = 0; k < RP_REFIT_LEVELS; ++k) {
        levels[k] = make_uint2(at, at + level_hist[k]);
        cursor[k] = at;
        at += level_hist[k];
    }
}
__global__ __launch_bounds__(256) void rp_k_lbvh_level_scatter(const RptrBvh4Node *nodes, int node_base, const int *count_ptr, uint32_t *cursor, uint32_t *list) {
    const int count = *count_ptr;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const uint32_t depth = nodes[node_base + k]._pad1[0];
        list[atomicAdd(&cursor[RP_REFIT_LEVELS - 1 - depth], 1u)] = (uint32_t)(node_base + k);
    }
}

// ------------------------------------------------------------------ refit by depth levels whose sizes live on the device
// one level of a mesh: the nodes list[levels[k].x .. levels[k].y)
__global__ __launch_bounds__(256) void rp_k_refit_level(RptrBvh4Node *nodes, float *node_box, const float *tri_box, const uint32_t *list, const uint2 *level) {
    const uint2 lv = *level;
    for (uint32_t i = lv.x + blockIdx.x * blockDim.x + threadIdx.x; i < lv.y; i += gridDim.x * blockDim.x) rp_refit_node(nodes, node_box, tri_box, nullptr, list[i]);
}
