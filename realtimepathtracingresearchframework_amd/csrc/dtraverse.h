// dtraverse.h -- two-level traversal of the compressed 4-wide BVH + Moeller-Trumbore for gfx950.
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
// instance-exit sentinel. Inner node: slab-test the four child boxes against
// [t_min, best_t] on the node's 8-bit grid (t = fma(q, step/d, (origin-o)/d));
// the hit children are ordered with three comparisons of their entry distances (closest-hit queries: before the clamp to t_min)
// (a miss counts as +inf; ties keep slot order): inside the pair of slots (0,1),
// inside the pair (2,3), and the pairs against each other by their nearer member.
// Continue with the first, push the others so that the first of the rest pops next.
// BLAS leaf: test all its triangles; TLAS leaf: transform the ray, push the
// sentinel, continue at the instance's root.
//
// The loop is "while-while": a wave stays in the node loop while any lane has an
// inner node, then handles leaves, so triangle code is not issued per node.
//
// Stack: the first RP_LDS_STACK entries of each lane live in LDS (column layout
// [level][thread]: conflict-free ds_read_b32/ds_write_b32), deeper entries
// spill to a per-thread slice of a global scratch buffer.
#pragma once
#include "bvh4.h"
#include "dshade.h"

#define RP_TRAVERSE_BLOCK 256
// 20 entries per lane in LDS (round 4; 24 until then): 20 KB per block let seven traversal blocks share a CU's 160 KB where 24 KB stopped at
// six -- which only matters since the kernels need 67-79 VGPRs. Deeper entries spill to the global scratch as before (the whole wave takes the
// generic path when a lane comes within three entries of the end). Measured (tools/ab.sh, one box): C2 1.143 / 1.150 -> 1.127 / 1.141 ms,
// C3 3.83 -> 3.71, C4 4.54 -> 4.48, two-level C4 6.54 -> 6.50; 16 entries: C3 3.67 but the two-level forest 6.68 (its stacks are deeper).
#ifndef RP_LDS_STACK
#define RP_LDS_STACK 20
#endif
// waves per SIMD the traversal kernels are compiled for (bounds the VGPR budget)
// (six since the end of round 4: without packed math the two-level instantiations need 76-80 VGPRs; two-level C4 6.70 / 6.68 -> 6.62 / 6.61 ms,
// two-level C3 4.38 -> 4.32. Rounds 1-3: five, 95-96 VGPRs.)
#ifndef RP_TRAVERSE_WAVES
#define RP_TRAVERSE_WAVES 6
#endif
#define RP_TRAVERSE_BOUNDS __launch_bounds__(RP_TRAVERSE_BLOCK, RP_TRAVERSE_WAVES)
// the shadow-ray kernels carry less per lane (no hit record to keep). The instantiation for scenes with ONE instance record and no alpha test
// (kernels.h rp_k_connect<., false, true>) is compiled for the register budget of EIGHT waves per SIMD: it uses 64 VGPRs without scratch
// (tools/kernel_regs.sh; round 4: 67 at a budget of 64 -- the hit record of round 5 carries two words less), and the LDS stacks (20 entries x
// 256 lanes x 4 bytes = 20 KB per block) let a CU hold seven such blocks: the host sizes a stand-alone launch's grid from that
// instantiation's own occupancy (rptr_hip.hip connect_blocks[1]). The two-level and the alpha-tested instantiations keep RP_TRAVERSE_WAVES
// (they spill 72-80 bytes at this budget: two-level C4 connect 2.5 -> 3.2 ms, profiles/r04_notes.md section 5); their grid (connect_blocks[0])
// comes from the two-level instantiation without alpha test -- an alpha-tested scene is launched with the same grid: persistent waves pull
// from one cursor, a few blocks more than fit only wait their turn.
#ifndef RP_CONNECT_WAVES
#define RP_CONNECT_WAVES 8
#endif
// After the build lost the SLP vectoriser and the packed slab test, the kernels for scenes with one instance record (no alpha test) need 77 /
// 71 / 67 VGPRs (first / later closest-hit, shadow rays). They are compiled for a register budget of SEVEN / EIGHT waves per SIMD (72 / 64
// VGPRs asked for, 79 / 71 / 67 used: the bound steers the scheduler's register / latency trade, the LDS stacks still limit a CU to six
// blocks): C2 1.196 / 1.196 / 1.190 -> 1.161 / 1.176 / 1.167 ms pipelined (-2.3 %), C4 4.66 -> 4.63, C3 3.95 -> 3.91 (tools/ab.sh, one box).
// The two-level and alpha-tested instantiations keep RP_TRAVERSE_WAVES (they need 96 VGPRs and would spill).
#ifndef RP_SINGLE_EXTEND_WAVES
#define RP_SINGLE_EXTEND_WAVES 7
#endif
// ... and the closest-hit launches of the later bounces (no camera-ray set-up in the refill): an experiment knob (default: as the first bounce)
#ifndef RP_EXTEND_LATER_WAVES
#define RP_EXTEND_LATER_WAVES RP_TRAVERSE_WAVES
#endif
#define RP_CONNECT_BOUNDS __launch_bounds__(RP_TRAVERSE_BLOCK, RP_CONNECT_WAVES)
// the node phase ends early when fewer than RP_NODE_MIN lanes are still at inner nodes and some lane waits with a leaf
#ifndef RP_NODE_MIN
#define RP_NODE_MIN 10
#endif
// ... and at least RP_LEAF_MIN lanes wait with a leaf
#ifndef RP_LEAF_MIN
#define RP_LEAF_MIN 1
#endif
#define RP_SENTINEL INT32_MIN
#define RP_EXIT (INT32_MIN + 1)

struct RpHitRec {
    float t, u, v;
    int tri;      // index of the hit triangle in RpScene::tris (its shading record: RpScene::shade[tri]; its primitive / geometry index: tris[tri])
    int inst_idx; // index into RpScene::insts (TLAS leaf order), -1 = miss
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
    return tnear <= tfar * 1.0000019f;
}

RP_DEV V3 rp_xform_point(const float4 r0, const float4 r1, const float4 r2, V3 p) {
    return v3(((r0.x * p.x + r0.y * p.y) + r0.z * p.z) + r0.w, ((r1.x * p.x + r1.y * p.y) + r1.z * p.z) + r1.w,
              ((r2.x * p.x + r2.y * p.y) + r2.z * p.z) + r2.w);
}
RP_DEV V3 rp_xform_dir(const float4 r0, const float4 r1, const float4 r2, V3 d) {
    return v3((r0.x * d.x + r0.y * d.y) + r0.z * d.z, (r1.x * d.x + r1.y * d.y) + r1.z * d.z, (r2.x * d.x + r2.y * d.y) + r2.z * d.z);
}

// ------------------------------------------------------------------ wave traversal engine
// Persistent waves with ray refill: a wave owns a pool of RP_FETCH queue entries
// (the first pool is assigned statically, later ones come from one shared
// cursor, one atomic per pool). Every lane runs an independent traversal state
// machine; when RP_REFILL_MIN or more lanes have finished their ray, the idle
// lanes take the next pool entries (ballot + mbcnt rank, no atomics), so short
// rays do not leave lanes idle while the longest ray of the wave finishes.
//   Load(i, o, d, tmin, tmax)  fetches queue entry i; false = the entry names no ray, nothing is traced or consumed for it
//   Done(i, hit)               consumes the result of entry i
//   Alpha(i, inst_idx, inst_id, geom, prim, u, v) -> true = ignore this candidate (ALPHA only: the reference's
//                              any-hit test of alpha-tested materials, pt_megakernel.glsl:153-212); called for the
//                              hits of triangles flagged RPTR_BVH_TRI_ALPHA that would otherwise be accepted
//
// The kernels are instruction-issue bound (profiles/r01_notes.md), so the node step
// is written for instruction count: packed v_pk_add/v_pk_mul for the 12 slab planes
// (operands pre-rotated into the (x,y) (z,x) (y,z) pairs in which the 12 floats of
// a node arrive), 32-bit node offsets (saddr loads), select instead of branch for
// the child choice, and leaf triangles fetched two at a time before testing.
// The node step's 24 plane distances: scalar fmas (default since round 4) or 12 v_pk_fma_f32 (rounds 1-3: -DRP_SLAB_PACKED=1). Same values
// either way. The packed form halves the instruction count, but v_pk_fma_f32 issues at half rate on gfx950 (tools/microbench/valu_issue.hip:
// two flops per issue slot both ways) and its register pairs cost VGPRs: scalar, the closest-hit kernels need 77 / 71 VGPRs instead of 77 / 76
// and the shadow-ray kernel 67 instead of 70, C2 1.228-1.242 -> 1.211-1.217 ms per frame, C4 4.79 -> 4.76 (profiles/r04_notes.md section 9).
#ifndef RP_SLAB_PACKED
#define RP_SLAB_PACKED 0
#endif
// Round 6, node step with fewer instructions (same values, same order of visits; 141 -> 131 VALU per step in the closest-hit kernels. Measured,
// profiles/r06_notes.md section 9: stand-alone closest-hit launches -2.5 %, one frame at a time -1.3 %, pipelined frames -0.6 % (C2) ... -1.8 % (C3)):
//   RP_STACK_BYTES the stack pointer is the LDS byte offset of the lane's next free entry (level * 1024 + 4 * thread) rather than the level
//   RP_ASM_MINMAX  the two min / max that take loop-carried operands (t_min, the best hit's t) are written as instructions: the compiler
//                 puts a v_max x, x in front of them to quiet a signalling NaN they cannot hold
//   RP_EXP_SDWA   2^exponent of the node's grid step with one v_lshlrev_b32_sdwa per axis (byte select + shift) instead of shift + and
#ifndef RP_NODE_TAIL_X2
#define RP_NODE_TAIL_X2 1
#endif
#ifndef RP_STACK_BYTES
#define RP_STACK_BYTES 1
#endif
#ifndef RP_ASM_MINMAX
#define RP_ASM_MINMAX 1
#endif
#ifndef RP_EXP_SDWA
#define RP_EXP_SDWA 1
#endif
#ifndef RP_REFILL_MIN
#define RP_REFILL_MIN 48
#endif
#ifndef RP_REFILL_MIN_ANY // the same thresholds for occlusion queries (tuned separately)
#define RP_REFILL_MIN_ANY RP_REFILL_MIN
#endif
#ifndef RP_NODE_MIN_ANY
#define RP_NODE_MIN_ANY RP_NODE_MIN
#endif
#ifndef RP_NODE_MIN_FIRST // ... and for the camera rays of the first bounce
#define RP_NODE_MIN_FIRST RP_NODE_MIN
#endif
#ifndef RP_REFILL_MIN_FIRST
#define RP_REFILL_MIN_FIRST RP_REFILL_MIN
#endif
#ifndef RP_SENTINEL_INLINE
#define RP_SENTINEL_INLINE 1
#endif
#ifndef RP_WORLD_INV_LDS // two-level scenes: the world-space ray's 1 / direction waits in LDS while the lane is inside an instance (0: recomputed on exit)
#define RP_WORLD_INV_LDS 1
#endif
#ifndef RP_FETCH_DIV
#define RP_FETCH_DIV 1u // a wave is dealt about 1/RP_FETCH_DIV of its fair share at a time
#endif
#ifndef RP_FETCH
#define RP_FETCH 256
#endif
#define RP_FETCH_DOC // queue entries a wave pulls per global atomic (upper bound; small launches take 64..256)

typedef float rp_f2 __attribute__((ext_vector_type(2)));
RP_DEV rp_f2 rp_mk2(float x, float y) { return rp_f2{x, y}; }

RP_DEV uint32_t rp_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

RP_DEV float rp_dot_fma(V3 a, V3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RP_DEV V3 rp_cross_fma(V3 a, V3 b) { return v3(fmaf(a.y, b.z, -(b.y * a.z)), fmaf(a.z, b.x, -(b.z * a.x)), fmaf(a.x, b.y, -(b.x * a.y))); }

#ifdef RP_PROF
__device__ unsigned long long rp_prof[16];
#endif
struct RpNoAlpha {
    RP_DEV bool operator()(uint32_t, int, int, int, int, float, float) const { return false; }
};
// LDSTOP (north_star: "LDS-staged node tiles"; RPTR_LDS_TOP=1, off by default): the first LDSTOP nodes of the node array -- trees are laid
// out breadth first, so these are the top levels, a third of all node visits on the height field for 64 nodes -- are copied into LDS when
// the block starts, and a node step reads nodes below that index from there (ds_read_b128) instead of through the vector cache. Measured
// (profiles/r03_notes.md section 8): no gain -- the top of the tree is what the L1 serves best (many lanes of a wave read the same line)
// -- which is why it is a separate instantiation that the default launches never touch.
#ifndef RP_LDS_TOP_NODES
#define RP_LDS_TOP_NODES 64
#endif
template <bool ANY, bool COUNT, int NODE_MIN = (ANY ? RP_NODE_MIN_ANY : RP_NODE_MIN), int REFILL_MIN = (ANY ? RP_REFILL_MIN_ANY : RP_REFILL_MIN),
          bool ALPHA = false, bool SINGLE = false, bool LOCAL = false, int LDSTOP = 0, bool EXTLDS = false, class Load, class Done, class Alpha>
RP_DEV void rp_wave_trace(const RpScene &sc, const uint32_t n, uint32_t *cursor, int *gstack, Load load, Done done, Alpha alpha,
                          uint32_t &n_nodes, uint32_t &n_tris, int *ext_stack = nullptr) {
    // EXTLDS: the caller owns the LDS part of the stacks (RP_LDS_STACK * RP_TRAVERSE_BLOCK ints) and shares it with its other phases
    // (kernels.h rp_k_tail: the closest-hit traversal, the shade scratch and the shadow-ray traversal of a block take turns on one arena)
    __shared__ int lds_stack_own[EXTLDS ? 1 : RP_LDS_STACK * RP_TRAVERSE_BLOCK];
    int *const lds_stack = EXTLDS ? ext_stack : lds_stack_own;
    // The two scheduling thresholds are the scene's (RpScene.node_min / refill_min, chosen at set_scene from the tree: rptr_hip.hip
    // traversal_preset; 0 = this instantiation's compile-time default). They decide WHEN a lane makes its next step, never the sequence of
    // steps of a ray: results and visit counts do not depend on them. Wave-uniform values: the compares below are scalar.
    const uint32_t node_min = sc.node_min > 0 ? (uint32_t)sc.node_min : (uint32_t)NODE_MIN;
    const uint32_t refill_min = sc.refill_min > 0 ? (uint32_t)sc.refill_min : (uint32_t)REFILL_MIN;
    const uint32_t tid = threadIdx.x;
    __shared__ float4 lds_top[LDSTOP > 0 ? LDSTOP * 4 : 1];
    if (LDSTOP > 0) { // stage the top of the tree (whole block; the caller's threads all get here)
        const uint32_t n_top = min((uint32_t)LDSTOP, sc.num_nodes);
        const float4 *src = reinterpret_cast<const float4 *>(sc.nodes);
        for (uint32_t k = threadIdx.x; k < n_top * 4u; k += blockDim.x) lds_top[k] = src[k];
        __syncthreads();
    }
    const uint32_t gstride = gridDim.x * blockDim.x;
    int *const glob = gstack + (blockIdx.x * blockDim.x + tid);
    const uint32_t lane = rp_lane_id();
    const unsigned long long lane_lt = (1ull << lane) - 1ull;
    // wave-uniform pool of queue entries
    // pool size: the queue is dealt out in 64..RP_FETCH entries per wave so that a short queue (late bounces) still
    // spreads over as many waves as it has 64-entry groups instead of 256 entries per wave on a quarter of the SIMDs
    // LOCAL: the queue belongs to this block alone (the tail kernel's block-local lists, kernels.h rp_k_tail): pools are dealt
    // to the block's own waves, `cursor` is a word in LDS
    const uint32_t nwaves = LOCAL ? (blockDim.x >> 6) : (gstride >> 6);
    const uint32_t fetch =
        LOCAL ? 64u : min(sc.fetch_max > 0 ? (uint32_t)sc.fetch_max : (uint32_t)RP_FETCH, max(64u, ((n + nwaves * RP_FETCH_DIV - 1u) / (nwaves * RP_FETCH_DIV) + 63u) & ~63u));
    uint32_t pool_next = ((LOCAL ? tid : (blockIdx.x * blockDim.x + tid)) >> 6) * fetch;
    uint32_t pool_end = min(n, pool_next + fetch);
    bool more = pool_next < n; // the shared cursor starts behind every static pool
    if (!more) return;
    // per-lane traversal state
    // the stack pointer counts in units of RP_SP_UNIT from sp_empty: entries (RP_STACK_BYTES 0) or LDS bytes (1: sp is the byte
    // offset of the lane's next free entry in the column layout, level * 1024 + 4 * thread; the level is sp >> 10 because 4 * thread < 1024,
    // and "level < L" is "sp < L * 1024" for the same reason)
#if RP_STACK_BYTES
    static_assert(RP_TRAVERSE_BLOCK * 4 == 1024, "RP_SP_LEVEL shifts by 10");
#define RP_SP_UNIT (RP_TRAVERSE_BLOCK * 4)
#define RP_SP_LEVEL(x) ((x) >> 10)
#define RP_SP_LDS(x) (*reinterpret_cast<int *>(reinterpret_cast<char *>(lds_stack) + (x)))
    const int sp_empty = (int)(tid * 4u);
#else
#define RP_SP_UNIT 1
#define RP_SP_LEVEL(x) (x)
#define RP_SP_LDS(x) (lds_stack[(x) * RP_TRAVERSE_BLOCK + tid])
    const int sp_empty = 0;
#endif
    int cur = RP_EXIT, sp = sp_empty;
    bool active = false; // lane holds a ray whose result has not been consumed yet
    uint32_t my_i = 0;
    V3 ro = v3s(0.f), rd = v3s(0.f), o = v3s(0.f), d = v3s(0.f);
    V3 inv = v3s(0.f); // 1/dir of the ray in the current space
    bool neg_x = false, neg_y = false, neg_z = false;
    float tmin = 0.f;
    RpHitRec best;
    best.t = 0.f;
    best.u = best.v = 0.f;
    best.tri = best.inst_idx = -1;
    int best_inst_id = -1, cur_inst = -1, cur_inst_id = -1;
    auto push = [&](int v) {
        if (sp < RP_LDS_STACK * RP_SP_UNIT)
            RP_SP_LDS(sp) = v;
        else
            glob[size_t(RP_SP_LEVEL(sp) - RP_LDS_STACK) * gstride] = v;
        sp += RP_SP_UNIT;
    };
    auto pop = [&]() -> int {
        sp -= RP_SP_UNIT;
        int v;
        if (sp < RP_LDS_STACK * RP_SP_UNIT)
            v = RP_SP_LDS(sp); // ds_read_b32; kept apart from the spill path so it is not a flat load
        else // an atomic (relaxed) load cannot be merged with the LDS read into one flat load
            v = __hip_atomic_load(glob + size_t(RP_SP_LEVEL(sp) - RP_LDS_STACK) * gstride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
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
    // Two-level scenes: 1 / direction of the WORLD-space ray, kept in LDS while the lane is inside an instance (3 KB per block) so that leaving
    // the instance is three LDS reads instead of three IEEE divisions -- the same bits. It matters because the exit is taken INSIDE the node
    // step (RP_SENTINEL_INLINE): on the instanced forest 12 % of a lane's node steps end with one (3.0 instance entries per ray, -DRP_PROF), so
    // nearly every wave iteration of the node phase pays for it.
    __shared__ float lds_world_inv[(!SINGLE && RP_WORLD_INV_LDS) ? 3 * RP_TRAVERSE_BLOCK : 1];
    auto leave_instance = [&]() { // the sentinel under an instance's entries: the query goes on in world space
        cur_inst = -1;
        cur_inst_id = -1;
        if (!SINGLE && RP_WORLD_INV_LDS) {
            o = ro;
            d = rd;
            inv = v3(lds_world_inv[tid], lds_world_inv[RP_TRAVERSE_BLOCK + tid], lds_world_inv[2 * RP_TRAVERSE_BLOCK + tid]);
            neg_x = __float_as_int(inv.x) < 0;
            neg_y = __float_as_int(inv.y) < 0;
            neg_z = __float_as_int(inv.z) < 0;
        } else
            set_ray(ro, rd);
        cur = pop();
    };
    // SINGLE (round 6): the one instance record of the scene -- three rows of world_to_object, the root of its tree, its id -- is read ONCE per
    // wave, into scalar registers, instead of by every refill: the waves are persistent, and five vector loads of a wave-uniform address
    // returned 3.5 KB per refill through the same L1 path as a node fetch (the same bits enter the same arithmetic).
#ifndef RP_SINGLE_HOIST
#define RP_SINGLE_HOIST 1
#endif
    float4 sw0 = make_float4(0.f, 0.f, 0.f, 0.f), sw1 = sw0, sw2 = sw0;
    int s_root = 0, s_inst_id = 0;
    if (SINGLE && RP_SINGLE_HOIST) {
        const float4 *ip = reinterpret_cast<const float4 *>(sc.insts);
        const float4 w0 = ip[0], w1 = ip[1], w2 = ip[2], meta = ip[3];
        auto uni = [](float x) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(x))); };
        sw0 = make_float4(uni(w0.x), uni(w0.y), uni(w0.z), uni(w0.w));
        sw1 = make_float4(uni(w1.x), uni(w1.y), uni(w1.z), uni(w1.w));
        sw2 = make_float4(uni(w2.x), uni(w2.y), uni(w2.z), uni(w2.w));
        s_root = __builtin_amdgcn_readfirstlane(__float_as_int(meta.x));
        s_inst_id = __builtin_amdgcn_readfirstlane(__float_as_int(meta.z));
    }
    const char *const node_base = reinterpret_cast<const char *>(sc.nodes);
    const char *const tri_base = reinterpret_cast<const char *>(sc.tris);
    const char *const inst_base = reinterpret_cast<const char *>(sc.insts);
    for (;;) {
#ifdef RP_PROF
        const long long prof_tr = wall_clock64();
#endif
        // ---- refill idle lanes
        const bool idle = cur == RP_EXIT;
        const unsigned long long idle_mask = __ballot(idle);
        const uint32_t nidle = (uint32_t)__popcll(idle_mask);
        if (nidle >= refill_min) {
            if (pool_next >= pool_end && more) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(cursor, fetch); // the cursor counts entries handed out behind the static pools
                base = __builtin_amdgcn_readfirstlane(base) + nwaves * fetch;
                if (base < n) {
                    pool_next = base;
                    pool_end = min(n, base + fetch);
                } else
                    more = false;
            }
            const uint32_t avail = pool_end - pool_next;
            if (avail > 0) {
                const uint32_t rank = (uint32_t)__popcll(idle_mask & lane_lt);
                if (idle && rank < avail) {
                    my_i = pool_next + rank;
                    float tmax;
                    if (load(my_i, ro, rd, tmin, tmax)) { // false: the entry names no ray (tile padding of the first queue): the lane stays idle
                    best.t = tmax;
                    best.u = best.v = 0.0f;
                    best.tri = best.inst_idx = -1;
                    best_inst_id = -1;
                    sp = sp_empty;
                    push(RP_EXIT);
                    if (SINGLE) {
                        // one instance record in the whole scene: the query starts inside it, at the root of its bottom-level
                        // tree -- no top-level node, no instance leaf and no sentinel to come back to (same arithmetic as
                        // entering the instance through its leaf; oracle/obvh.h traverse4 takes the same shortcut)
#if RP_SINGLE_HOIST
                        if (COUNT) n_nodes++; // the 64 bytes of the instance record, counted per query as the oracle's walk counts them
                        set_ray(rp_xform_point(sw0, sw1, sw2, ro), rp_xform_dir(sw0, sw1, sw2, rd));
                        cur_inst = 0;
                        cur_inst_id = s_inst_id;
                        cur = s_root;
#else
                        const float4 *ip = reinterpret_cast<const float4 *>(sc.insts);
                        const float4 w0 = ip[0], w1 = ip[1], w2 = ip[2], meta = ip[3];
                        if (COUNT) n_nodes++; // the 64 bytes of the instance record every query still reads
                        set_ray(rp_xform_point(w0, w1, w2, ro), rp_xform_dir(w0, w1, w2, rd));
                        cur_inst = 0;
                        cur_inst_id = __float_as_int(meta.z);
                        cur = __float_as_int(meta.x);
#endif
                    } else {
                        set_ray(ro, rd);
                        if (RP_WORLD_INV_LDS) {
                            lds_world_inv[tid] = inv.x;
                            lds_world_inv[RP_TRAVERSE_BLOCK + tid] = inv.y;
                            lds_world_inv[2 * RP_TRAVERSE_BLOCK + tid] = inv.z;
                        }
                        cur_inst = cur_inst_id = -1;
                        cur = 0;
                    }
                    active = true;
                    }
                }
                pool_next += min(nidle, avail);
            } else if (nidle == 64u)
                break; // nothing in flight, nothing left to fetch
        }
        // ---- inner nodes
#ifdef RP_PROF
        const long long prof_t0 = wall_clock64();
        if (lane == 0) atomicAdd(&rp_prof[8], (unsigned long long)(prof_t0 - prof_tr)); // refill section
        uint32_t prof_it = 0;
        const uint32_t prof_idle0 = (uint32_t)__popcll(__ballot(cur == RP_EXIT));
        const uint32_t prof_node0 = (uint32_t)__popcll(__ballot(cur >= 0));
#endif
        // node phase: keeps stepping while at least RP_NODE_MIN lanes are at an inner node (or nobody waits with a leaf)
        for (;;) {
            const unsigned long long want_node = __ballot(cur >= 0);
            if (want_node == 0ull) break;
#if RP_NODE_MIN > 1
            if ((uint32_t)__popcll(want_node) < node_min &&
                (uint32_t)__popcll(__ballot(cur < 0 && cur != RP_EXIT)) >= (uint32_t)RP_LEAF_MIN)
                break;
#endif
            if (cur >= 0) {
#ifdef RP_PROF
            prof_it++;
#endif
            // the whole wave takes the generic stack path when some lane is within 3 entries of the end of its LDS part
            const bool stack_slow = __any(sp >= (RP_LDS_STACK - 2) * RP_SP_UNIT);
#ifdef RP_PROF
            if (stack_slow && lane == 0) atomicAdd(&rp_prof[13], 1ull);
#endif
            int top = 0;
            if (!stack_slow) top = RP_SP_LDS(sp - RP_SP_UNIT); // read ahead: the item a miss would pop
            const char *np = node_base + (uint32_t(cur) << 6);
            float4 n0;  // origin.xyz, exp bytes
            uint4 n1;   // qlo.x qlo.y qlo.z qhi.x (4 children per dword)
            uint4 n2;   // qhi.y qhi.z child0 child1
            uint2 n3;   // child2 child3
            if (LDSTOP > 0 && uint32_t(cur) < (uint32_t)LDSTOP) {
                const float4 *lp = lds_top + (uint32_t(cur) << 2);
                n0 = lp[0];
                const float4 a1 = lp[1], a2 = lp[2], a3 = lp[3];
                n1 = make_uint4(__float_as_uint(a1.x), __float_as_uint(a1.y), __float_as_uint(a1.z), __float_as_uint(a1.w));
                n2 = make_uint4(__float_as_uint(a2.x), __float_as_uint(a2.y), __float_as_uint(a2.z), __float_as_uint(a2.w));
                n3 = make_uint2(__float_as_uint(a3.x), __float_as_uint(a3.y));
            } else {
                n0 = *reinterpret_cast<const float4 *>(np);
                n1 = *reinterpret_cast<const uint4 *>(np + 16);
                n2 = *reinterpret_cast<const uint4 *>(np + 32);
#if RP_NODE_TAIL_X2
                // child[2], child[3] as ONE 8-byte load the compiler can neither widen nor merge (a relaxed wave-scope atomic load is a plain
                // global_load_dwordx2): left alone it fetches bytes 48..63 with a fourth dwordx4, 8 bytes of padding per lane and node visit
                const unsigned long long c23 = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(np + 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                n3 = make_uint2((uint32_t)c23, (uint32_t)(c23 >> 32));
#else
                n3 = *reinterpret_cast<const uint2 *>(np + 48);
#endif
            }
            if (COUNT) n_nodes++;
            const uint32_t ex = __float_as_uint(n0.w);
            // plane distance t = q * A + B with A = step / d, B = (origin - o) / d
#if RP_EXP_SDWA
            uint32_t ex_x, ex_y, ex_z; // byte k of the word, shifted to the exponent field: the same bits as the shift + and below
            asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(ex_x) : "v"(ex));
            asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(ex_y) : "v"(ex));
            asm("v_lshlrev_b32_sdwa %0, 23, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(ex_z) : "v"(ex));
            const float ax = __uint_as_float(ex_x) * inv.x, ay = __uint_as_float(ex_y) * inv.y, az = __uint_as_float(ex_z) * inv.z;
#else
            const float ax = __uint_as_float((ex & 0xFFu) << 23) * inv.x, ay = __uint_as_float((ex & 0xFF00u) << 15) * inv.y,
                        az = __uint_as_float((ex & 0xFF0000u) << 7) * inv.z;
#endif
            const float bx = (n0.x - o.x) * inv.x, by = (n0.y - o.y) * inv.y, bz = (n0.z - o.z) * inv.z;
            // entry / exit planes by the sign of the direction (= min / max of the two plane distances, since
            // qlo <= qhi and the step is positive), selected once for the four children of a dword
            const uint32_t qnx = neg_x ? n1.w : n1.x, qfx = neg_x ? n1.x : n1.w;
            const uint32_t qny = neg_y ? n2.x : n1.y, qfy = neg_y ? n1.y : n2.x;
            const uint32_t qnz = neg_z ? n2.y : n1.z, qfz = neg_z ? n1.z : n2.y;
#if RP_SLAB_PACKED
            const rp_f2 ax2 = rp_mk2(ax, ax), ay2 = rp_mk2(ay, ay), az2 = rp_mk2(az, az);
            const rp_f2 bx2 = rp_mk2(bx, bx), by2 = rp_mk2(by, by), bz2 = rp_mk2(bz, bz);
#endif
            const float tfar_max = best.t;
            // a missed child becomes an empty slot with entry distance +inf: from here on "hit" is "ref != EMPTY" (an empty slot stays
            // one whatever its box says)
            int ref[4] = {(int)n2.z, (int)n2.w, (int)n3.x, (int)n3.y};
            float ent[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#if RP_SLAB_PACKED
                // (near, far) pairs: one packed fma per axis
                const rp_f2 tx = __builtin_elementwise_fma(rp_mk2((float)((qnx >> (8 * k)) & 0xFFu), (float)((qfx >> (8 * k)) & 0xFFu)), ax2, bx2);
                const rp_f2 ty = __builtin_elementwise_fma(rp_mk2((float)((qny >> (8 * k)) & 0xFFu), (float)((qfy >> (8 * k)) & 0xFFu)), ay2, by2);
                const rp_f2 tz = __builtin_elementwise_fma(rp_mk2((float)((qnz >> (8 * k)) & 0xFFu), (float)((qfz >> (8 * k)) & 0xFFu)), az2, bz2);
#else
                // the same six fmas as scalars (identical values): v_pk_fma_f32 issues at half rate on gfx950, and its register pairs cost moves
                const rp_f2 tx = rp_mk2(fmaf((float)((qnx >> (8 * k)) & 0xFFu), ax, bx), fmaf((float)((qfx >> (8 * k)) & 0xFFu), ax, bx));
                const rp_f2 ty = rp_mk2(fmaf((float)((qny >> (8 * k)) & 0xFFu), ay, by), fmaf((float)((qfy >> (8 * k)) & 0xFFu), ay, by));
                const rp_f2 tz = rp_mk2(fmaf((float)((qnz >> (8 * k)) & 0xFFu), az, bz), fmaf((float)((qfz >> (8 * k)) & 0xFFu), az, bz));
#endif
                // closest-hit queries order the children by the entry distance BEFORE it is clamped to t_min: a ray that starts inside several
                // overlapping boxes (instance boxes of a forest, secondary rays) has the same clamped entry distance for all of them and the
                // visit order would fall back to slot order -- which is right or wrong by the luck of the builder's left / right (37 or 46
                // node visits per ray on the instanced forest, depending on nothing but that). The unclamped value -- how far behind the
                // origin the box begins -- still tells them apart.
                const float tn_raw = fmaxf(fmaxf(tx.x, ty.x), tz.x);
#if RP_ASM_MINMAX
                // v_max_f32 / v_min_f32 as such: t_min and the best hit's t come around the loop in registers, and the compiler would first
                // quiet the signalling NaN they never are (v_max x, x: two instructions per node step); for every other operand value the
                // instruction IS fmaxf / fminf
                float tn, tfz;
                asm("v_max_f32 %0, %1, %2" : "=v"(tn) : "v"(tn_raw), "v"(tmin));
                asm("v_min_f32 %0, %1, %2" : "=v"(tfz) : "v"(tz.y), "v"(tfar_max));
                const float tf = fminf(fminf(tx.y, ty.y), tfz);
#else
                const float tn = fmaxf(tn_raw, tmin);
                const float tf = fminf(fminf(tx.y, ty.y), fminf(tz.y, tfar_max));
#endif
                // entry <= exit with a 1 + 2^-19 slack on the exit, as one fma: gap = entry - 1.0000019 exit <= 0. For an occlusion query the gap
                // is the order key as well: most negative first = the child the ray spends the longest stretch in, where an occluder is most
                // likely (flattened forest: 16.3 instead of 18.2 node visits, 3.8 instead of 5.7 triangle tests per shadow ray) -- for free.
                const float gap = fmaf(-1.0000019f, tf, tn);
                const bool hit = gap <= 0.0f;
                ref[k] = hit ? ref[k] : RPTR_BVH4_EMPTY;
                // (a missed child needs no +inf key in an occlusion query: its gap is positive, behind every hit child's)
                ent[k] = ANY ? gap : (hit ? tn_raw : INFINITY);
            }
            // front-to-back order with three comparisons instead of a sorting network over (key, payload) pairs: nearer first inside
            // each pair of slots, then the pair that holds the nearest child first. Against the full sort: +0.5 % node visits on the
            // 10 M-triangle forest, none on the height field (tools/order_probe.py); 18 VALU instructions fewer per node.
#define RP_SWAP_IF(c, i, j)                        \
    {                                              \
        const int ra_ = ref[i], rb_ = ref[j];      \
        ref[i] = (c) ? rb_ : ra_;                  \
        ref[j] = (c) ? ra_ : rb_;                  \
    }
            const bool sw_a = ent[1] < ent[0], sw_b = ent[3] < ent[2], sw_t = fminf(ent[2], ent[3]) < fminf(ent[0], ent[1]);
            RP_SWAP_IF(sw_a, 0, 1) RP_SWAP_IF(sw_b, 2, 3) RP_SWAP_IF(sw_t, 0, 2) RP_SWAP_IF(sw_t, 1, 3)
            const bool v0 = ref[0] != RPTR_BVH4_EMPTY, v1 = ref[1] != RPTR_BVH4_EMPTY, v2 = ref[2] != RPTR_BVH4_EMPTY, v3 = ref[3] != RPTR_BVH4_EMPTY;
#undef RP_SWAP_IF
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
                RP_SP_LDS(sp) = ref[3];
                sp += p3 ? RP_SP_UNIT : 0;
                RP_SP_LDS(sp) = ref[2];
                sp += p2 ? RP_SP_UNIT : 0;
                RP_SP_LDS(sp) = ref[1];
                sp += p1 ? RP_SP_UNIT : 0;
                nxt = v0 ? ref[0] : v1 ? ref[1] : v2 ? ref[2] : v3 ? ref[3] : top;
                sp -= (v0 || v1 || v2 || v3) ? 0 : RP_SP_UNIT;
            }
            cur = nxt;
            // the bottom-level tree is done: back to the top level right here (a few instructions for the lanes concerned) instead of
            // parking the lane until the wave's next leaf phase
            if (!SINGLE && RP_SENTINEL_INLINE && cur == RP_SENTINEL) leave_instance();
            }
        }
#ifdef RP_PROF
        {
            const long long prof_t1 = wall_clock64();
            uint32_t mx = prof_it;
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, off));
            uint32_t sm = prof_it;
            for (int off = 32; off > 0; off >>= 1) sm += (uint32_t)__shfl_xor((int)sm, off);
            if (lane == 0) {
                atomicAdd(&rp_prof[0], (unsigned long long)(prof_t1 - prof_t0)); // node-phase cycles (per wave)
                atomicAdd(&rp_prof[1], (unsigned long long)mx);                  // node-phase wave iterations
                atomicAdd(&rp_prof[2], (unsigned long long)sm);                  // node-phase lane iterations
                atomicAdd(&rp_prof[3], 1ull);                                    // phases
                atomicAdd(&rp_prof[5], (unsigned long long)mx * prof_idle0);                     // lane-iterations idle from the start of the phase
                atomicAdd(&rp_prof[6], (unsigned long long)mx * (64u - prof_idle0 - prof_node0)); // ... holding a leaf from the start
                atomicAdd(&rp_prof[7], (unsigned long long)mx * prof_node0 - sm);                // ... dropped out during the phase
            }
        }
        const long long prof_t2 = wall_clock64();
        {
            const bool isleaf = cur < 0 && cur != RP_EXIT;
            const uint32_t n_tri = (uint32_t)__popcll(__ballot(isleaf && cur_inst >= 0)), n_inst = (uint32_t)__popcll(__ballot(isleaf && cur_inst < 0));
            if (lane == 0) {
                atomicAdd(&rp_prof[9], (unsigned long long)n_tri);
                atomicAdd(&rp_prof[10], (unsigned long long)n_inst);
                atomicAdd(&rp_prof[11], (unsigned long long)(n_tri ? 1 : 0));
                atomicAdd(&rp_prof[12], (unsigned long long)(n_inst ? 1 : 0));
            }
        }
#endif
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
                    if (COUNT) n_nodes += 2; // 128-byte instance record
                    set_ray(rp_xform_point(qa0, qa1, qa2, ro), rp_xform_dir(qa0, qa1, qa2, rd));
                    cur_inst_id = __float_as_int(qb0.z);
                    push(RP_SENTINEL);
                    cur = __float_as_int(qb0.x);
                } else
                    cur = pop();
            } else {
                // BLAS leaf: canonical Moeller-Trumbore = oracle/obvh.h mt_intersect, same operations bit for bit
                bool any_hit = false;
                int tri_at = first; // index of triangle `qa` in RpScene::tris
#pragma unroll 1
                for (;;) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        if (j == 1 && (count < 2 || (ANY && any_hit))) break; // occlusion: the first accepted hit ends the query
                        const float4 q0 = j ? qb0 : qa0, q1 = j ? qb1 : qa1, q2 = j ? qb2 : qa2;
                        if (COUNT) n_tris++;
                        const V3 v0 = v3(q0.x, q0.y, q0.z), e1 = v3(q0.w, q1.x, q1.y), e2 = v3(q1.z, q1.w, q2.x);
                        const V3 p = rp_cross_fma(d, e2);
                        const float det = rp_dot_fma(e1, p);
                        const V3 tv = o - v0;
                        const float un = rp_dot_fma(tv, p);
                        const V3 q = rp_cross_fma(tv, e1);
                        const float vn = rp_dot_fma(d, q);
                        const float ad = fabsf(det);
                        // the sign bit of det (negative or -0; round 6: read as such -- a NaN determinant, whatever its sign, fails "ad > 0" below)
                        const bool neg = __float_as_int(det) < 0;
                        const float us = neg ? -un : un, vs = neg ? -vn : vn;
                        if (us >= 0.0f && vs >= 0.0f && us + vs <= ad && ad > 0.0f) {
                            const float inv_det = 1.0f / det;
                            const float t = rp_dot_fma(e2, q) * inv_det;
                            if (t > tmin) {
                                const int prim = __float_as_int(q2.y), geom = __float_as_int(q2.z);
                                // a triangle of a flattened scene names its own instance record (rptr_bvh.h), any other one
                                // belongs to the instance being traversed
                                const int tri_rec = (int)RPTR_BVH_TRI_INSTANCE(__float_as_uint(q2.w));
                                // (SINGLE: the instance being traversed is record 0 with the id read at the top, wave-uniform values)
                                const int in_inst = (SINGLE && RP_SINGLE_HOIST) ? 0 : cur_inst, in_inst_id = (SINGLE && RP_SINGLE_HOIST) ? s_inst_id : cur_inst_id;
                                const int hit_inst = tri_rec ? tri_rec : in_inst, hit_inst_id = tri_rec ? tri_rec - sc.flat_id_bias : in_inst_id;
                                bool accept = t < best.t;
                                if (!accept && t == best.t && best.inst_idx >= 0) { // a tie (rare): the smaller (instance, geometry, primitive) wins
                                    if (hit_inst_id != best_inst_id)
                                        accept = hit_inst_id < best_inst_id;
                                    else { // (the best hit's indices are not carried in registers: two words of its triangle record)
                                        const int *bt = reinterpret_cast<const int *>(tri_base + (size_t)(uint32_t)best.tri * 48u + 36u);
                                        const int best_prim = bt[0], best_geom = bt[1];
                                        accept = geom != best_geom ? geom < best_geom : prim < best_prim;
                                    }
                                }
                                if (ALPHA) {
                                    if (accept && (__float_as_uint(q2.w) & RPTR_BVH_TRI_ALPHA) != 0u)
                                        accept = !alpha(my_i, hit_inst, hit_inst_id, geom, prim, un * inv_det, vn * inv_det);
                                }
                                if (accept) {
                                    best.t = t;
                                    best.u = un * inv_det;
                                    best.v = vn * inv_det;
                                    best.tri = tri_at + j;
                                    best.inst_idx = hit_inst;
                                    best_inst_id = hit_inst_id;
                                    any_hit = true;
                                }
                            }
                        }
                    }
                    count -= 2;
                    if (count <= 0 || (ANY && any_hit)) break;
                    lp += 96; // leaves with more than two triangles: next pair
                    tri_at += 2;
                    qa0 = *reinterpret_cast<const float4 *>(lp);
                    qa1 = *reinterpret_cast<const float4 *>(lp + 16);
                    qa2 = *reinterpret_cast<const float4 *>(lp + 32);
                    qb0 = *reinterpret_cast<const float4 *>(lp + 48);
                    qb1 = *reinterpret_cast<const float4 *>(lp + 64);
                    qb2 = *reinterpret_cast<const float4 *>(lp + 80);
                }
                cur = (ANY && any_hit) ? RP_EXIT : pop();
                if (!SINGLE && RP_SENTINEL_INLINE && cur == RP_SENTINEL) leave_instance();
            }
        }
        if (active && cur == RP_EXIT) {
            done(my_i, best);
            active = false;
        }
#ifdef RP_PROF
        if (lane == 0) atomicAdd(&rp_prof[4], (unsigned long long)(wall_clock64() - prof_t2)); // leaf+done time (100 MHz ticks)
#endif
    }
}
