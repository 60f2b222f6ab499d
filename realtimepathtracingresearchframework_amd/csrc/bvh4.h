// bvh4.h -- encoding of the compressed 4-wide node (include/rptr_bvh.h), shared by the host builder
// (bvh_build.cpp) and the device refit kernels (kernels.h) so that "refit of unchanged vertices"
// reproduces the built tree bit for bit. Plain float arithmetic, no contraction (-ffp-contract=off).
#pragma once
#include "../../include/rptr_bvh.h"
#include <math.h>
#include <string.h>

#if defined(__HIP__)
#define RP_HD __host__ __device__ inline
#else
#define RP_HD inline
#endif

RP_HD uint32_t rp_bits_of(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    return u;
}
RP_HD float rp_float_of(uint32_t u) {
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

// exponent byte e with step = 2^(e-127) >= extent/254: 255 steps then cover the extent with 0.4 % to
// spare, which absorbs the roundings of origin + q*step
RP_HD uint32_t rp_bvh4_exponent(float extent) {
    const float x = extent / 254.0f;
    const uint32_t u = rp_bits_of(x) & 0x7FFFFFFFu;
    uint32_t e = u >> 23;
    if (u & 0x7FFFFFu) e += 1;
    if (e < 1u) e = 1u;     // flat / empty extents: the smallest normal step
    if (e > 253u) e = 253u; // keeps 2^(127-e) normal too
    return e;
}

struct RpBox4 { // float boxes of the (up to) four children of a node
    float lo[4][3], hi[4][3];
};

// One byte of RptrBvh4Node::order for two boxes (a "lower", b "upper" slot of a pair, or two pairs): the axis along which their centres
// are farthest apart (ties: the lowest axis) and the direction in which a ray meets b first. 0 when either is missing.
RP_HD uint32_t rp_bvh4_order_byte(const float alo[3], const float ahi[3], const float blo[3], const float bhi[3], bool a_valid, bool b_valid) {
    if (!a_valid || !b_valid) return 0u;
    int axis = 0;
    float best = -1.0f, diff_at = 0.0f;
    for (int a = 0; a < 3; ++a) {
        const float diff = (blo[a] + bhi[a]) - (alo[a] + ahi[a]); // twice the distance of the centres
        if (fabsf(diff) > best) {
            best = fabsf(diff);
            axis = a;
            diff_at = diff;
        }
    }
    // a is the lower one (or they coincide): b comes first for rays running in -axis; otherwise for rays running in +axis
    return diff_at >= 0.0f ? (1u << axis) : (8u << axis);
}

// Writes origin / exp / qlo / qhi / child / order of `out`; slots with child == RPTR_BVH4_EMPTY get the inverted box.
// node_lo/node_hi (optional) receive the exact float bounds of the node = union of its children.
RP_HD void rp_bvh4_encode(const RpBox4 &b, const int32_t child[4], RptrBvh4Node *out, float node_lo[3], float node_hi[3]) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int k = 0; k < 4; ++k) {
        if (child[k] == RPTR_BVH4_EMPTY) continue;
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], b.lo[k][a]);
            hi[a] = fmaxf(hi[a], b.hi[k][a]);
        }
    }
    if (node_lo)
        for (int a = 0; a < 3; ++a) {
            node_lo[a] = lo[a];
            node_hi[a] = hi[a];
        }
    RptrBvh4Node n;
    __builtin_memset(&n, 0, sizeof(n));
    for (int a = 0; a < 3; ++a) {
        const bool ok = lo[a] <= hi[a]; // false for a node without children
        const float org = ok ? lo[a] : 0.0f;
        const uint32_t e = rp_bvh4_exponent(ok ? hi[a] - lo[a] : 0.0f);
        const float step = rp_float_of(e << 23), inv_step = rp_float_of((254u - e) << 23);
        n.origin[a] = org;
        n.exp[a] = (uint8_t)e;
        for (int k = 0; k < 4; ++k) {
            if (child[k] == RPTR_BVH4_EMPTY) {
                n.qlo[a][k] = 255;
                n.qhi[a][k] = 0;
                continue;
            }
            float fl = floorf((b.lo[k][a] - org) * inv_step), fh = ceilf((b.hi[k][a] - org) * inv_step);
            int ql = (int)fminf(fmaxf(fl, 0.0f), 255.0f), qh = (int)fminf(fmaxf(fh, 0.0f), 255.0f);
            while (ql > 0 && org + (float)ql * step > b.lo[k][a]) --ql;   // the stored plane never lies inside the child
            while (qh < 255 && org + (float)qh * step < b.hi[k][a]) ++qh;
            n.qlo[a][k] = (uint8_t)ql;
            n.qhi[a][k] = (uint8_t)qh;
        }
    }
    for (int k = 0; k < 4; ++k) n.child[k] = child[k];
    {
        bool valid[4];
        for (int k = 0; k < 4; ++k) valid[k] = child[k] != RPTR_BVH4_EMPTY;
        float plo[2][3], phi[2][3]; // the two pairs as wholes
        for (int p = 0; p < 2; ++p)
            for (int a = 0; a < 3; ++a) {
                plo[p][a] = INFINITY;
                phi[p][a] = -INFINITY;
                for (int k = 2 * p; k < 2 * p + 2; ++k)
                    if (valid[k]) {
                        plo[p][a] = fminf(plo[p][a], b.lo[k][a]);
                        phi[p][a] = fmaxf(phi[p][a], b.hi[k][a]);
                    }
            }
        n.order = rp_bvh4_order_byte(plo[0], phi[0], plo[1], phi[1], valid[0] || valid[1], valid[2] || valid[3]) |
                  (rp_bvh4_order_byte(b.lo[0], b.hi[0], b.lo[1], b.hi[1], valid[0], valid[1]) << 8) |
                  (rp_bvh4_order_byte(b.lo[2], b.hi[2], b.lo[3], b.hi[3], valid[2], valid[3]) << 16);
    }
    *out = n;
}
