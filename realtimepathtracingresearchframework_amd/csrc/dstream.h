// dstream.h -- the traversal engine of the STREAMING frame (kernels.h rp_k_stream_trace): the node / leaf phases of dtraverse.h
// rp_wave_trace (same arithmetic, same visit order, same steps per ray: the marked region is GENERATED from that file by
// tools/gen_dstream.py out of tools/dstream_head.inc / dstream_tail.inc -- do not edit it by hand) around another life cycle of a lane.
//
// A lane of rp_wave_trace takes ONE ray from a queue whose length is known at launch. Here a lane takes an ITEM -- a path that left a
// shade: an optional shadow ray (occlusion query; its contribution is added to the path's radiance when the ray is unoccluded) followed by
// an optional continuation ray (closest-hit query, stored for the next shade) --, traces the item's rays one after the other (so the
// shadow ray's contribution is added before anything the next shade adds, as in the megakernel, with no cross-lane dependency), and
// pools of items arrive WHILE the kernel runs. The query kind is a per-lane run-time flag (`anyq`): four more selects per node step than
// the compile-time flag of rp_wave_trace.
//   Pool(first, end) -> 1: a pool of entries [first, end) for this wave; 0: none right now; -1: the frame is complete
//                       (wave-uniform; called when the wave's pool is empty and enough lanes are idle)
//   Begin(idx, ro, rd, tmin, tmax, anyq) -> false: the entry holds no item (padding): nothing is traced for it
//   Next(hit, ro, rd, tmin, tmax, anyq) -> consumes the result of the lane's ray; true: the item goes on with another ray
//   Report(round, idle): wave-uniform, between steps: lanes whose items ended since the last call are accounted for (round: a refill round
//                       has passed; idle: the wave has no ray in flight)
//   Alpha: as in rp_wave_trace
#pragma once
#include "dtraverse.h"

template <bool ALPHA, bool SINGLE, class Pool, class Begin, class Next, class Report, class Alpha>
RP_DEV void rp_wave_trace_items(const RpScene &sc, int *gstack, Pool pool, Begin begin, Next next, Report report, Alpha alpha) {
    __shared__ int lds_stack[RP_LDS_STACK * RP_TRAVERSE_BLOCK];
    const uint32_t node_min = sc.node_min > 0 ? (uint32_t)sc.node_min : (uint32_t)RP_NODE_MIN;
    const uint32_t refill_min = sc.refill_min > 0 ? (uint32_t)sc.refill_min : (uint32_t)RP_REFILL_MIN;
    const uint32_t tid = threadIdx.x;
    const uint32_t gstride = gridDim.x * blockDim.x;
    int *const glob = gstack + (blockIdx.x * blockDim.x + tid);
    const uint32_t lane = rp_lane_id();
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    uint32_t pool_next = 0, pool_end = 0; // wave-uniform
    uint32_t poll_wait = 0;               // iterations to let pass before the next look for a pool (after a look that found none)
    int cur = RP_EXIT, sp = 0;
    bool active = false, anyq = false;
    uint32_t my_i = 0;
    V3 ro = v3s(0.f), rd = v3s(0.f), o = v3s(0.f), d = v3s(0.f);
    V3 inv = v3s(0.f);
    bool neg_x = false, neg_y = false, neg_z = false;
    float tmin = 0.f;
    RpHitRec best;
    best.t = 0.f;
    best.u = best.v = 0.f;
    best.prim = best.inst_idx = best.geom = -1;
    int best_inst_id = -1, cur_inst = -1, cur_inst_id = -1;
    auto push = [&](int v) {
        if (sp < RP_LDS_STACK)
            lds_stack[sp * RP_TRAVERSE_BLOCK + tid] = v;
        else
            glob[size_t(sp - RP_LDS_STACK) * gstride] = v;
        ++sp;
    };
    auto pop = [&]() -> int {
        --sp;
        int v;
        if (sp < RP_LDS_STACK)
            v = lds_stack[sp * RP_TRAVERSE_BLOCK + tid];
        else
            v = __hip_atomic_load(glob + size_t(sp - RP_LDS_STACK) * gstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        return v;
    };
    auto set_ray = [&](V3 no, V3 nd) {
        o = no;
        d = nd;
        inv = v3(rp_safe_rcp(nd.x), rp_safe_rcp(nd.y), rp_safe_rcp(nd.z));
        neg_x = __float_as_int(inv.x) < 0;
        neg_y = __float_as_int(inv.y) < 0;
        neg_z = __float_as_int(inv.z) < 0;
    };
    auto leave_instance = [&]() {
        cur_inst = -1;
        cur_inst_id = -1;
        set_ray(ro, rd);
        cur = pop();
    };
    auto start = [&](float tmax) { // the lane's ray (ro, rd, tmin, tmax) begins: what rp_wave_trace does at a refill
        best.t = tmax;
        best.u = best.v = 0.0f;
        best.prim = best.geom = best.inst_idx = -1;
        best_inst_id = -1;
        sp = 0;
        push(RP_EXIT);
        if (SINGLE) {
            const float4 *ip = reinterpret_cast<const float4 *>(sc.insts);
            const float4 w0 = ip[0], w1 = ip[1], w2 = ip[2], meta = ip[3];
            set_ray(rp_xform_point(w0, w1, w2, ro), rp_xform_dir(w0, w1, w2, rd));
            cur_inst = 0;
            cur_inst_id = __float_as_int(meta.z);
            cur = __float_as_int(meta.x);
        } else {
            set_ray(ro, rd);
            cur_inst = cur_inst_id = -1;
            cur = 0;
        }
        active = true;
    };
    const char *const node_base = reinterpret_cast<const char *>(sc.nodes);
    const char *const tri_base = reinterpret_cast<const char *>(sc.tris);
    const char *const inst_base = reinterpret_cast<const char *>(sc.insts);
    uint32_t since_report = 0;
    for (;;) {
        // ---- account for ended items, refill idle lanes
        const bool idle = cur == RP_EXIT;
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t nidle = (uint32_t)__popcll(idle_mask);
        report(nidle >= refill_min || ++since_report >= 4u, nidle == 64u);
        if (nidle >= refill_min) {
            since_report = 0;
            if (pool_next >= pool_end) {
                if (poll_wait > 0u && nidle < 64u)
                    --poll_wait;
                else {
                    const int r = pool(pool_next, pool_end);
                    if (r < 0 && nidle == 64u) break;
                    if (r <= 0) {
                        pool_next = pool_end = 0;
                        if (nidle == 64u) {
                            __builtin_amdgcn_s_sleep(32);
                            continue;
                        }
                        poll_wait = 8u;
                    }
                }
            }
            const uint32_t avail = pool_end - pool_next;
            if (avail > 0) {
                const uint32_t rank = (uint32_t)__popcll(idle_mask & lane_lt);
                if (idle && rank < avail) {
                    my_i = pool_next + rank;
                    float tmax;
                    if (begin(my_i, ro, rd, tmin, tmax, anyq)) start(tmax);
                }
                pool_next += min(nidle, avail);
            }
        }
        // ======== generated from dtraverse.h (tools/gen_dstream.py): node phase, leaf phase
        // node phase: keeps stepping while at least RP_NODE_MIN lanes are at an inner node (or nobody waits with a leaf)
        for (;;) {
            const unsigned long long want_node = __ballot(cur >= 0);
            if (want_node == 0ull) break;
            if ((uint32_t)__popcll(want_node) < node_min &&
                (uint32_t)__popcll(__ballot(cur < 0 && cur != RP_EXIT)) >= (uint32_t)RP_LEAF_MIN)
                break;
            if (cur >= 0) {
            // the whole wave takes the generic stack path when some lane is within 3 entries of the end of its LDS part
            const bool stack_slow = __any(sp > RP_LDS_STACK - 3);
            int top = 0;
            if (!stack_slow) top = lds_stack[(sp - 1) * RP_TRAVERSE_BLOCK + tid]; // read ahead: the item a miss would pop
            const char *np = node_base + (uint32_t(cur) << 6);
            float4 n0;  // origin.xyz, exp bytes
            uint4 n1;   // qlo.x qlo.y qlo.z qhi.x (4 children per dword)
            uint4 n2;   // qhi.y qhi.z child0 child1
            uint2 n3;   // child2 child3
            n0 = *reinterpret_cast<const float4 *>(np);
            n1 = *reinterpret_cast<const uint4 *>(np + 16);
            n2 = *reinterpret_cast<const uint4 *>(np + 32);
            n3 = *reinterpret_cast<const uint2 *>(np + 48);
            
            const uint32_t ex = __float_as_uint(n0.w);
            // plane distance t = q * A + B with A = step / d, B = (origin - o) / d
            const float ax = __uint_as_float((ex & 0xFFu) << 23) * inv.x, ay = __uint_as_float((ex & 0xFF00u) << 15) * inv.y,
                        az = __uint_as_float((ex & 0xFF0000u) << 7) * inv.z;
            const float bx = (n0.x - o.x) * inv.x, by = (n0.y - o.y) * inv.y, bz = (n0.z - o.z) * inv.z;
            // entry / exit planes by the sign of the direction (= min / max of the two plane distances, since
            // qlo <= qhi and the step is positive), selected once for the four children of a dword
            const uint32_t qnx = neg_x ? n1.w : n1.x, qfx = neg_x ? n1.x : n1.w;
            const uint32_t qny = neg_y ? n2.x : n1.y, qfy = neg_y ? n1.y : n2.x;
            const uint32_t qnz = neg_z ? n2.y : n1.z, qfz = neg_z ? n1.z : n2.y;
            const float tfar_max = best.t;
            // a missed child becomes an empty slot with entry distance +inf: from here on "hit" is "ref != EMPTY" (an empty slot stays
            // one whatever its box says)
            int ref[4] = {(int)n2.z, (int)n2.w, (int)n3.x, (int)n3.y};
            float ent[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the same six fmas as scalars (identical values): v_pk_fma_f32 issues at half rate on gfx950, and its register pairs cost moves
                const rp_f2 tx = rp_mk2(fmaf((float)((qnx >> (8 * k)) & 0xFFu), ax, bx), fmaf((float)((qfx >> (8 * k)) & 0xFFu), ax, bx));
                const rp_f2 ty = rp_mk2(fmaf((float)((qny >> (8 * k)) & 0xFFu), ay, by), fmaf((float)((qfy >> (8 * k)) & 0xFFu), ay, by));
                const rp_f2 tz = rp_mk2(fmaf((float)((qnz >> (8 * k)) & 0xFFu), az, bz), fmaf((float)((qfz >> (8 * k)) & 0xFFu), az, bz));
                // closest-hit queries order the children by the entry distance BEFORE it is clamped to t_min: a ray that starts inside several
                // overlapping boxes (instance boxes of a forest, secondary rays) has the same clamped entry distance for all of them and the
                // visit order would fall back to slot order -- which is right or wrong by the luck of the builder's left / right (37 or 46
                // node visits per ray on the instanced forest, depending on nothing but that). The unclamped value -- how far behind the
                // origin the box begins -- still tells them apart.
                const float tn_raw = fmaxf(fmaxf(tx.x, ty.x), tz.x);
                const float tn = fmaxf(tn_raw, tmin);
                const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tfar_max));
                // entry <= exit with a 1 + 2^-19 slack on the exit, as one fma: gap = entry - 1.0000019 exit <= 0. For an occlusion query the gap
                // is the order key as well: most negative first = the child the ray spends the longest stretch in, where an occluder is most
                // likely (flattened forest: 16.3 instead of 18.2 node visits, 3.8 instead of 5.7 triangle tests per shadow ray) -- for free.
                const float gap = fmaf(-1.0000019f, tf, tn);
                const bool hit = gap <= 0.0f;
                ref[k] = hit ? ref[k] : RPTR_BVH4_EMPTY;
                // (a missed child needs no +inf key in an occlusion query: its gap is positive, behind every hit child's)
                ent[k] = anyq ? gap : (hit ? tn_raw : INFINITY);
            }
            // front-to-back order with three comparisons instead of a sorting network over (key, payload) pairs: nearer first inside
            // each pair of slots, then the pair that holds the nearest child first. Against the full sort: +0.5 % node visits on the
            // 10 M-triangle forest, none on the height field (tools/order_probe.py); 18 VALU instructions fewer per node.
            const bool sw_a = ent[1] < ent[0], sw_b = ent[3] < ent[2], sw_t = fminf(ent[2], ent[3]) < fminf(ent[0], ent[1]);
#define RP_SWAP_IF(c, i, j)                        \
    {                                              \
        const int ra_ = ref[i], rb_ = ref[j];      \
        ref[i] = (c) ? rb_ : ra_;                  \
        ref[j] = (c) ? ra_ : rb_;                  \
    }
            RP_SWAP_IF(sw_a, 0, 1) RP_SWAP_IF(sw_b, 2, 3) RP_SWAP_IF(sw_t, 0, 2) RP_SWAP_IF(sw_t, 1, 3)
#undef RP_SWAP_IF
            const bool v0 = ref[0] != RPTR_BVH4_EMPTY, v1 = ref[1] != RPTR_BVH4_EMPTY, v2 = ref[2] != RPTR_BVH4_EMPTY, v3 = ref[3] != RPTR_BVH4_EMPTY;
            // the first hit in that order is next; the later ones go on the stack, farthest first
            const bool p3 = v3 && (v0 || v1 || v2), p2 = v2 && (v0 || v1), p1 = v1 && v0;
            int nxt;
            if (__builtin_expect(stack_slow, 0)) { // rare: some lane is about to leave the LDS part of its stack
                if (p3) push(ref[3]);
                if (p2) push(ref[2]);
                if (p1) push(ref[1]);
                nxt = v0 ? ref[0] : v1 ? ref[1] : v2 ? ref[2] : v3 ? ref[3] : pop();
            } else { // branch-free: write, then advance only for real entries; no child hit = no push, and the entry read ahead
                     // from the top of the stack is the next item
                lds_stack[sp * RP_TRAVERSE_BLOCK + tid] = ref[3];
                sp += p3 ? 1 : 0;
                lds_stack[sp * RP_TRAVERSE_BLOCK + tid] = ref[2];
                sp += p2 ? 1 : 0;
                lds_stack[sp * RP_TRAVERSE_BLOCK + tid] = ref[1];
                sp += p1 ? 1 : 0;
                nxt = v0 ? ref[0] : v1 ? ref[1] : v2 ? ref[2] : v3 ? ref[3] : top;
                sp -= (v0 || v1 || v2 || v3) ? 0 : 1;
            }
            cur = nxt;
            // the bottom-level tree is done: back to the top level right here (a few instructions for the lanes concerned) instead of
            // parking the lane until the wave's next leaf phase
            if (!SINGLE && RP_SENTINEL_INLINE && cur == RP_SENTINEL) leave_instance();
            }
        }
        // ---- one leaf / sentinel item. A BLAS leaf (two triangles = 96 bytes) and a TLAS leaf (the first 64 bytes of
        // an instance record) are fetched by the same six loads, so that a phase with both kinds costs one round trip.
        if (!SINGLE && cur == RP_SENTINEL) {
            leave_instance();
        } else if (cur < 0 && cur != RP_EXIT) {
            const int first = RPTR_BVH_LEAF_FIRST(cur);
            int count = RPTR_BVH_LEAF_COUNT(cur);
            const bool is_inst = !SINGLE && cur_inst < 0;
            const char *lp = is_inst ? inst_base + (size_t)(uint32_t)first * sizeof(RptrBvhInstance) : tri_base + (size_t)(uint32_t)first * 48u;
            float4 qa0 = *reinterpret_cast<const float4 *>(lp), qa1 = *reinterpret_cast<const float4 *>(lp + 16),
                   qa2 = *reinterpret_cast<const float4 *>(lp + 32), qb0 = *reinterpret_cast<const float4 *>(lp + 48),
                   qb1 = *reinterpret_cast<const float4 *>(lp + 64), qb2 = *reinterpret_cast<const float4 *>(lp + 80);
            if (is_inst) {
                // TLAS leaf: enter the instance (rows of world_to_object, then blas_root / geometry_base / instance_id / flags)
                if (count > 0) {
                    cur_inst = first;
                    
                    set_ray(rp_xform_point(qa0, qa1, qa2, ro), rp_xform_dir(qa0, qa1, qa2, rd));
                    cur_inst_id = __float_as_int(qb0.z);
                    push(RP_SENTINEL);
                    cur = __float_as_int(qb0.x);
                } else
                    cur = pop();
            } else {
                // BLAS leaf: canonical Moeller-Trumbore = oracle/obvh.h mt_intersect, same operations bit for bit
                bool any_hit = false;
#pragma unroll 1
                for (;;) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j == 1 && (count < 2 || (anyq && any_hit))) break; // occlusion: the first accepted hit ends the query
                        const float4 q0 = j ? qb0 : qa0, q1 = j ? qb1 : qa1, q2 = j ? qb2 : qa2;
                        
                        const V3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q0.w, q1.x, q1.y), e2 = v3(q1.z, q1.w, q2.x);
                        const V3 p = rp_cross_fma(d, e2);
                        const float det = rp_dot_fma(e1, p);
                        const V3 tv = o - v0;
                        const float un = rp_dot_fma(tv, p);
                        const V3 q = rp_cross_fma(tv, e1);
                        const float vn = rp_dot_fma(d, q);
                        const float ad = fabsf(det);
                        const bool neg = det < 0.0f || (det == 0.0f && __float_as_int(det) < 0);
                        const float us = neg ? -un : un, vs = neg ? -vn : vn;
                        if (us >= 0.0f && vs >= 0.0f && us + vs <= ad && ad > 0.0f) {
                            const float inv_det = 1.0f / det;
                            const float t = rp_dot_fma(e2, q) * inv_det;
                            if (t > tmin) {
                                const int prim = __float_as_int(q2.y), geom = __float_as_int(q2.z);
                                // a triangle of a flattened scene names its own instance record (rptr_bvh.h), any other one
                                // belongs to the instance being traversed
                                const int tri_rec = (int)RPTR_BVH_TRI_INSTANCE(__float_as_uint(q2.w));
                                const int hit_inst = tri_rec ? tri_rec : cur_inst, hit_inst_id = tri_rec ? tri_rec - 1 : cur_inst_id;
                                bool accept = t < best.t;
                                if (!accept && t == best.t && best.inst_idx >= 0) {
                                    if (hit_inst_id != best_inst_id)
                                        accept = hit_inst_id < best_inst_id;
                                    else if (geom != best.geom)
                                        accept = geom < best.geom;
                                    else
                                        accept = prim < best.prim;
                                }
                                if (ALPHA) {
                                    if (accept && (__float_as_uint(q2.w) & RPTR_BVH_TRI_ALPHA) != 0u)
                                        accept = !alpha(my_i, hit_inst, hit_inst_id, geom, prim, un * inv_det, vn * inv_det);
                                }
                                if (accept) {
                                    best.t = t;
                                    best.u = un * inv_det;
                                    best.v = vn * inv_det;
                                    best.prim = prim;
                                    best.geom = geom;
                                    best.inst_idx = hit_inst;
                                    best_inst_id = hit_inst_id;
                                    any_hit = true;
                                }
                            }
                        }
                    }
                    count -= 2;
                    if (count <= 0 || (anyq && any_hit)) break;
                    lp += 96; // leaves with more than two triangles: next pair
                    qa0 = *reinterpret_cast<const float4 *>(lp);
                    qa1 = *reinterpret_cast<const float4 *>(lp + 16);
                    qa2 = *reinterpret_cast<const float4 *>(lp + 32);
                    qb0 = *reinterpret_cast<const float4 *>(lp + 48);
                    qb1 = *reinterpret_cast<const float4 *>(lp + 64);
                    qb2 = *reinterpret_cast<const float4 *>(lp + 80);
                }
                cur = (anyq && any_hit) ? RP_EXIT : pop();
                if (!SINGLE && RP_SENTINEL_INLINE && cur == RP_SENTINEL) leave_instance();
            }
        }
        // ======== end of the generated region
        if (active && cur == RP_EXIT) {
            float tmax;
            if (next(best, ro, rd, tmin, tmax, anyq))
                start(tmax);
            else
                active = false;
        }
    }
}
