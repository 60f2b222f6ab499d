// dtraverse.h -- two-level BVH2 traversal + Moeller-Trumbore for gfx950.
//
// Replaces what the reference gets from the Vulkan driver / RT hardware:
// rayQueryProceedEXT loops at vulkan/pt_megakernel.glsl:440-475 (closest hit),
// :216-272 (visibility) and vulkan/rt_intersect.comp:46-51. Query semantics
// (DESIGN.md "Ray query semantics"): hit accepted for t_min < t < t_max,
// barycentrics weight vertex 1 and 2 (rendering/rt/hit.glsl:70), closest hit =
// smallest t with ties resolved to the smallest (instance, geometry, primitive)
// triple, which makes the result independent of tree shape and visit order.
//
// Visit order (it defines the node/triangle counts of the roofline model; the
// CPU oracle walks the exported tree in exactly this order): work items are
// inner nodes (>= 0), packed leaves (<= -2, include/rptr_bvh.h) and the
// instance-exit sentinel. Inner node: slab-test both child boxes against
// [t_min, best_t]; both hit -> push the farther, continue with the nearer
// (tie -> child 0); BLAS leaf: test all its triangles; TLAS leaf: transform the
// ray, push the sentinel, continue at the instance's root.
//
// The loop is "while-while": a wave stays in the node loop while any lane has an
// inner node, then handles leaves, so triangle code is not issued per node.
//
// Stack: the first RP_LDS_STACK entries of each lane live in LDS (column layout
// [level][thread]: conflict-free ds_read_b32/ds_write_b32), deeper entries
// spill to a per-thread slice of a global scratch buffer.
#pragma once
#include "dshade.h"

#define RP_TRAVERSE_BLOCK 256
#ifndef RP_LDS_STACK
#define RP_LDS_STACK 24
#endif
#define RP_SENTINEL INT32_MIN
#define RP_EXIT (INT32_MIN + 1)

struct RpHitRec {
    float t, u, v;
    int prim;
    int inst_idx; // index into RpScene::insts (TLAS leaf order), -1 = miss
    int geom;     // geometry index inside the mesh
};

struct RpStack {
    int *lds;        // &lds_stack[threadIdx.x]
    int *glob;       // &global_stack[global thread id]
    uint32_t gstride;
    int sp;
    RP_DEV void push(int v) {
        if (sp < RP_LDS_STACK)
            lds[sp * RP_TRAVERSE_BLOCK] = v;
        else
            glob[size_t(sp - RP_LDS_STACK) * gstride] = v;
        ++sp;
    }
    RP_DEV int pop() {
        --sp;
        return sp < RP_LDS_STACK ? lds[sp * RP_TRAVERSE_BLOCK] : glob[size_t(sp - RP_LDS_STACK) * gstride];
    }
};

RP_DEV float rp_safe_rcp(float x) { return fabsf(x) >= 1e-30f ? 1.0f / x : copysignf(1e30f, x); }

RP_DEV bool rp_slab(V3 lo, V3 hi, V3 o, V3 id, float tmin, float tmax, float &tnear) {
    float t0x = (lo.x - o.x) * id.x, t1x = (hi.x - o.x) * id.x;
    float t0y = (lo.y - o.y) * id.y, t1y = (hi.y - o.y) * id.y;
    float t0z = (lo.z - o.z) * id.z, t1z = (hi.z - o.z) * id.z;
    tnear = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tfar = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tnear <= tfar * 1.0000005f;
}

RP_DEV V3 rp_xform_point(const float4 r0, const float4 r1, const float4 r2, V3 p) {
    return v3(((r0.x * p.x + r0.y * p.y) + r0.z * p.z) + r0.w, ((r1.x * p.x + r1.y * p.y) + r1.z * p.z) + r1.w,
              ((r2.x * p.x + r2.y * p.y) + r2.z * p.z) + r2.w);
}
RP_DEV V3 rp_xform_dir(const float4 r0, const float4 r1, const float4 r2, V3 d) {
    return v3((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z, (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
}

template <bool ANY, bool COUNT>
RP_DEV bool rp_traverse(const RpScene &sc, V3 ro, V3 rd, float tmin, float tmax, RpHitRec &best, RpStack &st, uint32_t &n_nodes,
                        uint32_t &n_tris) {
    best.t = tmax;
    best.u = best.v = 0.0f;
    best.prim = -1;
    best.inst_idx = -1;
    best.geom = -1;
    int best_inst_id = -1;
    st.sp = 0;
    st.push(RP_EXIT);
    V3 o = ro, d = rd;
    V3 id = v3(rp_safe_rcp(d.x), rp_safe_rcp(d.y), rp_safe_rcp(d.z));
    int cur_inst = -1, cur_inst_id = -1;
    int cur = 0;
    for (;;) {
        // ---- inner nodes
        while (cur >= 0) {
            const float4 *np = reinterpret_cast<const float4 *>(sc.nodes + cur);
            const float4 a = np[0], b = np[1], c = np[2];
            const int4 k = *reinterpret_cast<const int4 *>(np + 3);
            if (COUNT) n_nodes++;
            float tn0, tn1;
            const bool h0 = rp_slab(v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), o, id, tmin, best.t, tn0);
            const bool h1 = rp_slab(v3(b.z, b.w, c.x), v3(c.y, c.z, c.w), o, id, tmin, best.t, tn1);
            if (h0 && h1) {
                const bool near1 = tn1 < tn0;
                st.push(near1 ? k.x : k.y);
                cur = near1 ? k.y : k.x;
            } else if (h0)
                cur = k.x;
            else if (h1)
                cur = k.y;
            else
                cur = st.pop();
        }
        if (cur == RP_EXIT) break;
        if (cur == RP_SENTINEL) {
            cur_inst = -1;
            cur_inst_id = -1;
            o = ro;
            d = rd;
            id = v3(rp_safe_rcp(d.x), rp_safe_rcp(d.y), rp_safe_rcp(d.z));
            cur = st.pop();
            continue;
        }
        const int first = RPTR_BVH_LEAF_FIRST(cur), count = RPTR_BVH_LEAF_COUNT(cur);
        if (cur_inst < 0) {
            // ---- TLAS leaf: enter the instance
            if (count > 0) {
                cur_inst = first;
                const float4 *ip = reinterpret_cast<const float4 *>(sc.insts + cur_inst);
                const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
                const int4 meta = *reinterpret_cast<const int4 *>(ip + 6);
                if (COUNT) n_nodes += 2; // 128-byte instance record
                o = rp_xform_point(r0, r1, r2, ro);
                d = rp_xform_dir(r0, r1, r2, rd);
                id = v3(rp_safe_rcp(d.x), rp_safe_rcp(d.y), rp_safe_rcp(d.z));
                cur_inst_id = meta.z;
                st.push(RP_SENTINEL);
                cur = meta.x;
            } else
                cur = st.pop();
            continue;
        }
        // ---- BLAS leaf
        for (int i = 0; i < count; ++i) {
            const float4 *tp = reinterpret_cast<const float4 *>(sc.tris + (first + i));
            const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
            if (COUNT) n_tris++;
            const V3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q0.w, q1.x, q1.y), e2 = v3(q1.z, q1.w, q2.x);
            const int prim = __float_as_int(q2.y), geom = __float_as_int(q2.z);
            // canonical Moeller-Trumbore (operation order = oracle/obvh.h mt_intersect)
            const V3 p = cross3(d, e2);
            const float det = dot3(e1, p);
            if (det == 0.0f) continue;
            const float inv = 1.0f / det;
            const V3 tv = o - v0;
            const float u = dot3(tv, p) * inv;
            if (!(u >= 0.0f && u <= 1.0f)) continue;
            const V3 q = cross3(tv, e1);
            const float v = dot3(d, q) * inv;
            if (!(v >= 0.0f && u + v <= 1.0f)) continue;
            const float t = dot3(e2, q) * inv;
            if (!(t > tmin)) continue;
            bool accept = t < best.t;
            if (!accept && t == best.t && best.inst_idx >= 0) {
                if (cur_inst_id != best_inst_id)
                    accept = cur_inst_id < best_inst_id;
                else if (geom != best.geom)
                    accept = geom < best.geom;
                else
                    accept = prim < best.prim;
            }
            if (!accept) continue;
            best.t = t;
            best.u = u;
            best.v = v;
            best.prim = prim;
            best.geom = geom;
            best.inst_idx = cur_inst;
            best_inst_id = cur_inst_id;
            if (ANY) return true;
        }
        cur = st.pop();
    }
    return best.inst_idx >= 0;
}
