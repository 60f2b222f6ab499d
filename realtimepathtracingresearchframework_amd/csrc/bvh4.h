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

// Writes origin / exp / qlo / qhi / child of `out`; slots with child == RPTR_BVH4_EMPTY get the inverted box.
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
    *out = n;
}
