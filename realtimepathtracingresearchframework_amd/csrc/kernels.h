// kernels.h -- the wavefront stages that replace the reference's megakernel.
//
//   raygen   pt_megakernel.glsl:310-325   camera rays, RNG seeding: computed by the first extend / shade (rp_primary_ray)
//   extend   pt_megakernel.glsl:440-475   closest-hit queries      (persistent waves)
//   sort     (new)                        regroup hit paths by material id
//   shade    pt_megakernel.glsl:480-730   miss/sky, hit attributes, emitter MIS, NEE
//                                         sampling, BSDF sampling, Russian roulette
//   connect  pt_megakernel.glsl:216-272   shadow queries + NEE accumulation (persistent)
//   resolve  accumulate.glsl:44-74 + process_samples.comp:69-200
//   trace    rt_intersect.comp:31-68      RQ_CLOSEST batch query
//
// Path state is SoA, indexed by path id = sample_slot * npix_padded + tiled pixel
// slot. Every kernel reads its work count from device memory, so one frame is a
// fixed launch sequence without host round trips.
//
// Queue discipline (one device word sustains only ~88 atomics/us on MI355X, see
// MI355X_MICROARCH.md "dequeue"): producers compact into an LDS staging buffer
// with wave-level ballot + mbcnt prefix sums and publish a whole 1024-entry chunk
// with ONE global atomic; persistent consumers pull 256 entries per atomic.
#pragma once
#include "dtraverse.h"
#include "dstream.h"
#include <type_traits>

// Round 4 (less path-state traffic, every value exact): the continuation ray's t_min is a function of its origin and the path length
// (rp_geometry_scale_to_tmin: the extend recomputes it) and its t_max is a constant, so the two .w words carry the path length and the
// generator instead and the separate 8-byte (generator, path length) array is gone; the shadow ray of a vertex starts where the
// continuation ray does, with the same offset, so ray_o serves both (RpShadowRays has no origin array): a first-bounce path writes 96
// bytes where it wrote 120, a later shade reads 88 instead of 96.
struct RpPathState {
    float4 *ray_o;   // origin.xyz, total_t (the path length up to the origin): origin of the continuation ray AND of the vertex's shadow ray
    float4 *ray_d;   // dir.xyz, bits(rng state)
    float4 *thr;     // throughput.xyz, prev_bounce_pdf
    float4 *illum;   // illum.xyz, bits(bounce)
    float4 *footprint; // the texture footprint (2 x 2, column major) of paths through scenes with textures (else NULL)
    uint32_t *alpha_rng; // the alpha-test generator of closest-hit queries when the point set is not the uniform one (else NULL)
    float4 *hit_tuv; // t, u, v, bits(prim)
    int2 *hit_ids;   // inst_idx, geom
};
// shadow rays live at the slot of their path (at most one per path and bounce);
// the shadow queue itself only carries path ids
struct RpShadowRays {
    float4 *d;       // dir.xyz, t_max (the origin and t_min: RpPathState.ray_o)
    float4 *contrib; // radiance to add if visible .xyz
    uint32_t *ids;   // compacted path ids
};
// device-side counters, one block of them per frame
// queue heads of one bounce. They live in per-bounce slots that one memset per frame zeroes, so nothing has to be
// reset between the launches of a frame (a reset kernel per bounce was 9 launches and ~32 us per frame).
struct RpBounceCounters {
    uint32_t queue_count;    // rays this bounce extends (written by raygen / the previous bounce's shade)
    uint32_t shadow_count;   // shadow rays this bounce's shade emitted
    uint32_t cursor_extend;  // entries handed out behind the static first pools (dtraverse.h)
    uint32_t cursor_connect;
};
#define RP_MAX_BOUNCES 64 // RptrRenderParams.max_path_depth is validated against it
struct RpCounters {
    RpBounceCounters bounce[RP_MAX_BOUNCES + 1];
    unsigned long long rays_closest, rays_shadow, nodes, tris, hits_shaded, nodes_shadow, tris_shadow;
    uint32_t stack_overflow;
    uint32_t _pad2;
};

#define RP_CHUNK 1024     // entries a producer block publishes per global atomic

// ---- path state that crosses workgroups INSIDE a launch (rp_k_frame below; SH = true)
// Per-XCD L2s are not coherent with each other and a CU's vector L1 is never refreshed by another CU's stores (MI355X_MICROARCH.md
// "inter-workgroup visibility"): what one block writes and another block of the same launch reads goes through device-coherent accesses on
// both sides -- `sc0 sc1` buffer loads / stores (loads bypass the L1, stores write through the L2; a 16-byte sc1 access costs what a plain
// one does, and path state is read once per bounce, so nothing is lost in the L1) -- plus a drained store queue before the entry that
// names the path is published. SH = false (the stand-alone stages: a kernel boundary lies between producer and consumer) compiles to the
// plain accesses it always was.
typedef uint32_t rp_u4v __attribute__((ext_vector_type(4)));
typedef uint32_t rp_u2v __attribute__((ext_vector_type(2)));
#ifndef RP_AUX_COHERENT
#define RP_AUX_COHERENT 17 // sc0 | sc1
#endif
RP_DEV __amdgpu_buffer_rsrc_t rp_rsrc(const void *base) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0x7FFFFFFF, 0x00020000); }
template <bool SH>
RP_DEV float4 rp_ld4(const float4 *a, uint32_t i) {
    if (!SH) return a[i];
    const rp_u4v v = __builtin_amdgcn_raw_buffer_load_b128(rp_rsrc(a), int(i * 16u), 0, RP_AUX_COHERENT);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <bool SH>
RP_DEV void rp_st4(float4 *a, uint32_t i, float4 x) {
    if (!SH) {
        a[i] = x;
        return;
    }
    const rp_u4v v = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rp_rsrc(a), int(i * 16u), 0, RP_AUX_COHERENT);
}
template <bool SH>
RP_DEV float2 rp_ld2(const float2 *a, uint32_t i) {
    if (!SH) return a[i];
    const rp_u2v v = __builtin_amdgcn_raw_buffer_load_b64(rp_rsrc(a), int(i * 8u), 0, RP_AUX_COHERENT);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
template <bool SH>
RP_DEV void rp_st2(float2 *a, uint32_t i, float2 x) {
    if (!SH) {
        a[i] = x;
        return;
    }
    const rp_u2v v = {__float_as_uint(x.x), __float_as_uint(x.y)};
    __builtin_amdgcn_raw_buffer_store_b64(v, rp_rsrc(a), int(i * 8u), 0, RP_AUX_COHERENT);
}
template <bool SH>
RP_DEV uint32_t rp_ld1(const uint32_t *a, uint32_t i) { // i: element index
    if (!SH) return a[i];
    return __builtin_amdgcn_raw_buffer_load_b32(rp_rsrc(a), int(i * 4u), 0, RP_AUX_COHERENT);
}
template <bool SH>
RP_DEV void rp_st1(uint32_t *a, uint32_t i, uint32_t x) {
    if (!SH) {
        a[i] = x;
        return;
    }
    __builtin_amdgcn_raw_buffer_store_b32(x, rp_rsrc(a), int(i * 4u), 0, RP_AUX_COHERENT);
}

// ---- wave64 helpers
// reserves one slot per flagged lane with one atomic per wave (counter may live in LDS or global memory)
RP_DEV uint32_t rp_wave_append(uint32_t *counter, bool flag) {
    const unsigned long long mask = __ballot(flag);
    if (mask == 0ull) return 0u;
    const uint32_t lane = rp_lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if (int(lane) == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}
RP_DEV uint32_t rp_wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
// publishes `n_local` staged ids from LDS to a global queue: one atomic per block
RP_DEV void rp_block_flush(const uint32_t *staged, uint32_t n_local, uint32_t *queue, uint32_t *counter, uint32_t *s_base) {
    if (threadIdx.x == 0) *s_base = n_local ? atomicAdd(counter, n_local) : 0u;
    __syncthreads();
    const uint32_t base = *s_base;
    for (uint32_t j = threadIdx.x; j < n_local; j += blockDim.x) queue[base + j] = staged[j];
}

// ------------------------------------------------------------------ raygen
// The camera ray of path p (pt_megakernel.glsl:314-325 + :330-352 pinhole branch): a pure function of the frame
// constants and the path id, so nothing of it is stored -- the first extend and the first shade both call it
// (saves writing and re-reading 72 bytes of path state per pixel sample). Returns false for padding slots.
// origin: the camera position of the path's frame; du_dv (may be NULL): its image-plane axes. Frames with cameras of their own
// (f.per_frame_cams; wave-uniform, so the branch is a scalar one) read cams[frame] per lane -- a launch sequence's frames differ by sample
// slot, and a wave's pool may straddle two of them -- in every instantiation (round 4: routing them through the general TABLE kernels
// cost 3-5 % of a C2 frame for what is three cached 16-byte loads per camera ray).
template <bool TABLE>
RP_DEV bool rp_primary_ray_ex(const RpFrame &f, uint32_t p, RpRng &rng, V3 &dir, int &lx, int &ly, uint32_t &sslot, V3 &origin, V3 *du_dv = nullptr) {
    sslot = rp_div(p, f.div_npix_padded);
    const uint32_t slot = p - sslot * uint32_t(f.npix_padded);
    lx = ly = 0;
    if (!rp_slot_to_local(f, slot, lx, ly)) return false;
    const int gy = rp_local_row_to_global(f, ly);
    if (gy >= f.height) return false;
    const RpSlotFrame sf = rp_slot_frame(f, sslot);
    rng = rp_rng_open<TABLE>(f, sf, uint32_t(lx), uint32_t(gy));
    V2 point = v2(float(lx) + 0.5f, float(gy) + 0.5f);
    // (enable_raster_taa != 0 is rendered by the TABLE instantiation: the shipped path has no branch on it)
    if (!TABLE || f.rp.enable_raster_taa == 0) point = point + (rp_draw2<TABLE>(f, rng, 0u /* DIM_PIXEL_X */) - v2(0.5f, 0.5f));
    point = v2(point.x / float(f.width), point.y / float(f.height));
    if (TABLE && f.rp.enable_raster_taa != 0) point = point + rp_screen_jitter(f, sf.frame_offset, sf.frame_id) * 0.5f; // pt_megakernel.glsl:319-320
    V3 du = ld3(f.cam_du), dv = ld3(f.cam_dv), tl = ld3(f.cam_dir_top_left);
    origin = ld3(f.cam_pos);
    if (f.per_frame_cams != 0) {
        const RpCam &c = f.cams[min(sf.frame, (uint32_t)(RP_BATCH_CAMS - 1))];
        du = ld3(c.du);
        dv = ld3(c.dv);
        tl = ld3(c.dir_top_left);
        origin = ld3(c.pos);
    }
    if (du_dv) {
        du_dv[0] = du;
        du_dv[1] = dv;
    }
    dir = norm3(point.x * du + point.y * dv + tl);
    return true;
}
// the generator of path p at a later bounce: index / pixel recomputed, the state `s` from the path state
template <bool TABLE>
RP_DEV RpRng rp_rng_resume(const RpFrame &f, uint32_t p, uint32_t s) {
    RpRng r;
    r.s = s;
    r.index = r.pix = 0u;
    if (TABLE && f.rng_variant != RPTR_RNG_VARIANT_UNIFORM) {
        const uint32_t sslot = rp_div(p, f.div_npix_padded);
        int lx = 0, ly = 0;
        (void)rp_slot_to_local(f, p - sslot * uint32_t(f.npix_padded), lx, ly);
        r = rp_rng_open<TABLE>(f, rp_slot_frame(f, sslot), uint32_t(lx), uint32_t(rp_local_row_to_global(f, ly)));
        r.s = s;
    }
    return r;
}
// the alpha-test generator of closest-hit queries: the path's own for the uniform point set (`#define alpha_rng rng`), a separate LCG
// seeded like the uniform one otherwise (pt_megakernel.glsl:354-358)
RP_DEV uint32_t rp_alpha_seed(const RpFrame &f, uint32_t p) {
    const uint32_t sslot = rp_div(p, f.div_npix_padded);
    int lx = 0, ly = 0;
    (void)rp_slot_to_local(f, p - sslot * uint32_t(f.npix_padded), lx, ly);
    const RpSlotFrame sf = rp_slot_frame(f, sslot);
    return rp_rng_seed(sf.sample_index, sf.frame_offset, uint32_t(lx), uint32_t(rp_local_row_to_global(f, ly)), uint32_t(f.width));
}
template <bool TABLE>
RP_DEV bool rp_primary_ray(const RpFrame &f, uint32_t p, RpRng &rng, V3 &dir, V3 &origin) {
    int lx, ly;
    uint32_t sslot;
    return rp_primary_ray_ex<TABLE>(f, p, rng, dir, lx, ly, sslot, origin);
}

// ------------------------------------------------------------------ extend (closest hit), persistent waves
// FIRST: bounce 0, the rays are the camera rays (computed, not loaded)
// ALPHA: the scene has alpha-tested materials. The test of a candidate may draw from the path's generator
// (pt_megakernel.glsl:354-358), so the lane carries it through the traversal and hands it back in the path state.
// SINGLE: the scene has one instance record; queries start inside it (dtraverse.h).
// LOCAL: `queue` / `cursor` are a block-local list and its cursor in LDS (rp_k_tail).
// SH / EXTLDS (rp_k_frame): path state through device-coherent accesses; the stacks' LDS belongs to the caller. base: the path id of entry 0
// when the queue is the identity (FIRST, queue == NULL)
template <bool COUNT, bool FIRST, bool ALPHA, bool SINGLE, bool LOCAL, bool TABLE, int LDSTOP = 0, bool SH = false, bool EXTLDS = false>
RP_DEV void rp_extend_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const uint32_t *queue, uint32_t n, uint32_t *cursor, RpCounters *ctr,
                           int *gstack, uint32_t base = 0u, int *ext_stack = nullptr) {
    uint32_t n_nodes = 0, n_tris = 0;
    uint32_t lane_rng = 0, lane_rng_in = 0; // ALPHA only
    uint32_t lane_p = 0; // the path whose ray this lane traces
    auto load = [&](uint32_t i, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool {
        const uint32_t p = (FIRST && !queue) ? base + i : queue[i]; // FIRST: the first queue is the identity (NULL) unless the caller stored it
        lane_p = p;
        if (FIRST) {
            RpRng rng;
            rd = v3(0.0f, 0.0f, 1.0f);
            if (!rp_primary_ray<TABLE>(f, p, rng, rd, ro)) { // tile padding: no such pixel sample. A miss is recorded (the regrouping pass reads it)
                ps.hit_ids[p] = make_int2(-1, -1);
                return false;
            }
            tmin = 0.0f;
            tmax = 2.e32f;
            if (ALPHA) lane_rng = (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) ? rng.s : rp_alpha_seed(f, p);
        } else {
            const float4 o = rp_ld4<SH>(ps.ray_o, p), d = rp_ld4<SH>(ps.ray_d, p);
            ro = xyz(o);
            rd = xyz(d);
            tmin = rp_geometry_scale_to_tmin(ro, o.w); // (what the shade computed for its shadow ray: same operands, same bits)
            tmax = 1e20f;
            if (ALPHA)
                lane_rng = lane_rng_in = (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) ? __float_as_uint(d.w) : rp_ld1<SH>(ps.alpha_rng, p);
        }
        return true;
    };
    auto done = [&](uint32_t, const RpHitRec &h) {
        const uint32_t p = lane_p;
        ps.hit_tuv[p] = make_float4(h.t, h.u, h.v, __int_as_float(h.prim));
        ps.hit_ids[p] = make_int2(h.inst_idx, h.geom);
        if (ALPHA) {
            if (TABLE && f.rng_variant != RPTR_RNG_VARIANT_UNIFORM) {
                if (FIRST || lane_rng != lane_rng_in) rp_st1<SH>(ps.alpha_rng, p, lane_rng);
            } else if (FIRST)
                rp_st1<SH>(reinterpret_cast<uint32_t *>(ps.ray_d), 4u * p + 3u, lane_rng); // the first shade takes it from here (f.alpha_test)
            else if (lane_rng != lane_rng_in)
                rp_st1<SH>(reinterpret_cast<uint32_t *>(ps.ray_d), 4u * p + 3u, lane_rng);
        }
    };
    auto alpha = [&](uint32_t, int inst_idx, int, int geom, int prim, float u, float v) -> bool {
        return rp_alpha_rejects(sc, inst_idx, geom, prim, u, v, lane_rng);
    };
    rp_wave_trace<false, COUNT, (FIRST ? RP_NODE_MIN_FIRST : RP_NODE_MIN), (FIRST ? RP_REFILL_MIN_FIRST : RP_REFILL_MIN), ALPHA, SINGLE, LOCAL, LDSTOP, EXTLDS>(
        sc, n, cursor, gstack, load, done, alpha, n_nodes, n_tris, ext_stack);
    if (COUNT) {
        n_nodes = rp_wave_sum_u32(n_nodes);
        n_tris = rp_wave_sum_u32(n_tris);
        if (rp_lane_id() == 0) {
            atomicAdd(&ctr->nodes, (unsigned long long)n_nodes);
            atomicAdd(&ctr->tris, (unsigned long long)n_tris);
        }
    }
}
template <bool COUNT, bool FIRST, bool ALPHA, bool SINGLE, bool TABLE>
__global__ __launch_bounds__(RP_TRAVERSE_BLOCK, ((SINGLE && !ALPHA) ? RP_SINGLE_EXTEND_WAVES : (FIRST ? RP_TRAVERSE_WAVES : RP_EXTEND_LATER_WAVES))) void rp_k_extend(RpScene sc, RpFrame f, RpPathState ps, const uint32_t *queue, RpBounceCounters *bc, RpCounters *ctr,
                                               int *gstack) {
    rp_extend_body<COUNT, FIRST, ALPHA, SINGLE, false, TABLE>(sc, f, ps, queue, bc->queue_count, &bc->cursor_extend, ctr, gstack);
}
// the same with the top of the tree staged in LDS (dtraverse.h LDSTOP; RPTR_LDS_TOP=1): plain scenes only (one instance record, no alpha
// test, the LCG point set)
template <bool FIRST>
__global__ RP_TRAVERSE_BOUNDS void rp_k_extend_ldstop(RpScene sc, RpFrame f, RpPathState ps, const uint32_t *queue, RpBounceCounters *bc, RpCounters *ctr,
                                                      int *gstack) {
    rp_extend_body<false, FIRST, false, true, false, false, RP_LDS_TOP_NODES>(sc, f, ps, queue, bc->queue_count, &bc->cursor_extend, ctr, gstack);
}

// ------------------------------------------------------------------ connect (shadow rays), persistent waves
// ALPHA: shadow rays test alpha-tested candidates with a generator seeded per candidate from (primitive ^ frame_id,
// instance ^ frame_offset, pixel), pt_megakernel.glsl:251-262 -- independent of the order in which candidates turn up.
// ids: the compacted path ids of the shadow rays (sq.ids, or the tail kernel's block-local list: LOCAL)
template <bool COUNT, bool ALPHA, bool SINGLE, bool LOCAL, int LDSTOP = 0, bool SH = false, bool EXTLDS = false>
RP_DEV void rp_connect_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq, const uint32_t *ids, uint32_t n, uint32_t *cursor,
                            RpCounters *ctr, int *gstack, int *ext_stack = nullptr) {
    uint32_t n_nodes = 0, n_tris = 0;
    auto alpha = [&](uint32_t i, int inst_idx, int inst_id, int geom, int prim, float u, float v) -> bool {
        const uint32_t p = ids[i];
        const uint32_t sslot = rp_div(p, f.div_npix_padded);
        const uint32_t slot = p - sslot * uint32_t(f.npix_padded);
        int lx = 0, ly = 0;
        (void)rp_slot_to_local(f, slot, lx, ly);
        const int gy = rp_local_row_to_global(f, ly);
        const RpSlotFrame sf = rp_slot_frame(f, sslot);
        uint32_t rng = rp_rng_seed(uint32_t(prim) ^ sf.frame_id, uint32_t(inst_id) ^ sf.frame_offset, uint32_t(lx), uint32_t(gy), uint32_t(f.width));
        return rp_alpha_rejects(sc, inst_idx, geom, prim, u, v, rng);
    };
    auto load = [&](uint32_t i, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool {
        const uint32_t p = ids[i];
        const float4 o = rp_ld4<SH>(ps.ray_o, p), d = sq.d[p]; // the vertex: origin of the shadow ray and of the continuation ray
        ro = xyz(o);
        rd = xyz(d);
        tmin = rp_geometry_scale_to_tmin(ro, o.w);
        tmax = d.w;
        return true;
    };
    auto done = [&](uint32_t i, const RpHitRec &h) {
        if (h.inst_idx < 0) { // visible: NEE contribution arrives (nee.glsl:76-84)
            const uint32_t p = ids[i];
            const float4 c = sq.contrib[p];
            float4 il = rp_ld4<SH>(ps.illum, p);
            il.x += c.x;
            il.y += c.y;
            il.z += c.z;
            rp_st4<SH>(ps.illum, p, il);
        }
    };
    rp_wave_trace<true, COUNT, RP_NODE_MIN_ANY, RP_REFILL_MIN_ANY, ALPHA, SINGLE, LOCAL, LDSTOP, EXTLDS>(sc, n, cursor, gstack, load, done, alpha, n_nodes, n_tris,
                                                                                                        ext_stack);
    if (COUNT) {
        n_nodes = rp_wave_sum_u32(n_nodes);
        n_tris = rp_wave_sum_u32(n_tris);
        if (rp_lane_id() == 0) {
            atomicAdd(&ctr->nodes_shadow, (unsigned long long)n_nodes);
            atomicAdd(&ctr->tris_shadow, (unsigned long long)n_tris);
        }
    }
}
// (six waves per SIMD only for scenes with one instance record: the two-level walk keeps the instance's state alive -- 72-80 bytes of scratch
// at 80 VGPRs, two-level C4 connect 2.5 -> 3.2 ms: measured, so those instantiations keep the closest-hit kernels' bound)
template <bool COUNT, bool ALPHA, bool SINGLE>
__global__ __launch_bounds__(RP_TRAVERSE_BLOCK, ((SINGLE && !ALPHA) ? RP_CONNECT_WAVES : RP_TRAVERSE_WAVES)) void rp_k_connect(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpBounceCounters *bc, RpCounters *ctr, int *gstack) {
    rp_connect_body<COUNT, ALPHA, SINGLE, false>(sc, f, ps, sq, sq.ids, bc->shadow_count, &bc->cursor_connect, ctr, gstack);
}
template <int LDSTOP>
__global__ RP_TRAVERSE_BOUNDS void rp_k_connect_ldstop(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpBounceCounters *bc, RpCounters *ctr, int *gstack) {
    rp_connect_body<false, false, true, false, LDSTOP>(sc, f, ps, sq, sq.ids, bc->shadow_count, &bc->cursor_connect, ctr, gstack);
}

// ------------------------------------------------------------------ shade
#ifndef RP_SHADE_WAVES
#define RP_SHADE_WAVES 4 // minimum waves per SIMD the shade kernels are compiled for (bounds their VGPR budget)
#endif
// LIGHTS = false: the scene has no emissive triangles, every NEE sample goes to the sun (sun_radiance.w == 1,
// vulkan/render_sky.cpp:68-71) and the binned-RIS code is compiled out (fewer registers, smaller kernel)
// TEX = false: no material of the scene reads a texture (textured parameters, normal maps): sampling code compiled out
// LOCAL (rp_k_tail): `order` is a block-local list of n <= RP_CHUNK path ids; the survivors and the shadow rays are not
// published to the global queues but left in shared memory for the caller (local_next / local_shadow, counts in n_next / n_shadow)
// RP_KERNARG_RELOAD: left alone, the shade kernels keep ~300 scalar values of their by-value arguments (RpFrame, RpScene) alive across the
// whole loop and spill 200-330 of them into VGPR lanes (v_writelane / v_readlane: VALU instructions, a tenth of the loop's). The loop body
// therefore reads the two structs from the kernel-argument segment again at three points -- scalar loads from constant memory behind an asm
// barrier that keeps them from being hoisted -- so that nothing of them has to stay in registers from one stretch to the next: 25 spilled
// SGPRs instead of 199-330, VGPR spills 44 -> 5 (glTF + lights) and 114 -> 33 (textured), the Lambert kernel 110 -> 97 VGPRs; shade launches
// -6 % on C3 and on textured scenes, frames -1 % (C2) ... -3 % (C3) (profiles/r03_notes.md section 9). -DRP_KERNARG_RELOAD=0: the old code.
#ifndef RP_KERNARG_RELOAD
#define RP_KERNARG_RELOAD 1
#endif
template <class T>
RP_DEV const T &rp_kernarg(uint32_t offset) {
    typedef const T __attribute__((address_space(4))) *KP;
    uint64_t a = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr() + offset;
    asm volatile("" : "+s"(a));
    return *(const T *)(KP)a;
}
// both callers (rp_k_shade, rp_k_tail) start their argument lists with (RpScene sc, RpFrame f, ...): explicit kernel arguments lie in the
// kernel-argument segment from offset 0 at their natural alignment (the whole-frame parity tests would not survive a wrong offset)
#define RP_KERNARG_OFF_SC 0u
#define RP_KERNARG_OFF_F ((uint32_t)((sizeof(RpScene) + alignof(RpFrame) - 1) / alignof(RpFrame) * alignof(RpFrame)))
#if RP_KERNARG_RELOAD
#define RP_RELOAD_ARGS                                               \
    const RpFrame &f = rp_kernarg<RpFrame>(RP_KERNARG_OFF_F);        \
    const RpScene &sc = rp_kernarg<RpScene>(RP_KERNARG_OFF_SC);      \
    (void)f;                                                         \
    (void)sc;
#else
#define RP_RELOAD_ARGS
#endif
// the LDS buffers of rp_shade_body when the caller owns them (EXTLDS; rp_k_frame): `next` and `shadow` (RP_CHUNK words each) outlive the call
// (the survivors and the shadow rays of the chunk), `list` (RP_CHUNK words), `ris_req` (2048 floats) and `ris_contrib` (4096 floats; LIGHTS
// only) are scratch the caller may reuse between calls
struct RpShadeLds {
    uint32_t *next, *shadow, *list;
    float *ris_req, *ris_contrib;
};
#define RP_SHADE_RIS_REQ_FLOATS ((256 / 64) * 64 * 8)
#define RP_SHADE_RIS_CONTRIB_FLOATS ((256 / 64) * 64 * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE)
// SH / EXTLDS / base: as for rp_extend_body
// STREAM (rp_k_stream_shade): the hit records were written by ANOTHER workgroup (read around the L1), and the chunk leaves ONE list of
// items -- path id | RP_ITEM_CONT (the path goes on) | RP_ITEM_SHADOW (a shadow ray is pending) -- in `next` instead of two lists
#define RP_ITEM_CONT 0x40000000u
#define RP_ITEM_SHADOW 0x80000000u
#define RP_ITEM_PATH 0x3FFFFFFFu
#define RP_ITEM_NONE 0xFFFFFFFFu // padding of a sealed chunk: no item
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX, bool LOCAL, bool TABLE, bool SH = false, bool EXTLDS = false, bool STREAM = false>
RP_DEV void rp_shade_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq, const uint32_t *order, const uint32_t n,
                          uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count, RpCounters *ctr, uint32_t *&local_next, uint32_t &n_next,
                          uint32_t *&local_shadow, uint32_t &n_shadow, uint32_t base = 0u, const RpShadeLds *ext = nullptr) {
    __shared__ uint32_t s_next_own[EXTLDS ? 1 : RP_CHUNK], s_shadow_own[EXTLDS ? 1 : RP_CHUNK];
    __shared__ uint32_t s_nn, s_ns, s_base;
    __shared__ uint32_t s_stat[3];
    __shared__ uint32_t s_list_own[EXTLDS ? 1 : RP_CHUNK]; // path ids of the chunk, hits first
    __shared__ uint32_t s_nhit, s_nmiss;
    // per wave: the tri-light requests of its lanes (hit point, normal, bin) and the contributions of their bins
    __shared__ float s_ris_req_own[LIGHTS && !EXTLDS ? RP_SHADE_RIS_REQ_FLOATS : 1];
    __shared__ float s_ris_contrib_own[LIGHTS && !EXTLDS ? RP_SHADE_RIS_CONTRIB_FLOATS : 1];
    uint32_t *const s_next = EXTLDS ? ext->next : s_next_own, *const s_shadow = EXTLDS ? ext->shadow : s_shadow_own;
    uint32_t *const s_list = EXTLDS ? ext->list : s_list_own;
    float *const s_ris_req = EXTLDS ? ext->ris_req : s_ris_req_own, *const s_ris_contrib = EXTLDS ? ext->ris_contrib : s_ris_contrib_own;
    if (threadIdx.x < 3) s_stat[threadIdx.x] = 0;
    if (LOCAL) {
        if (threadIdx.x == 0) {
            s_nn = 0;
            s_ns = 0;
        }
        __syncthreads(); // the body runs once per bounce in the tail kernel: the previous call's readers are done
    }
    const uint32_t nchunks = (n + RP_CHUNK - 1) / RP_CHUNK;
    uint32_t my_closest = 0, my_shadow = 0, my_hits = 0;
    for (uint32_t chunk = LOCAL ? 0u : blockIdx.x; chunk < nchunks; chunk += LOCAL ? nchunks : gridDim.x) {
        if (threadIdx.x == 0) {
            s_nn = 0;
            s_ns = 0;
            s_nhit = 0;
            s_nmiss = 0;
        }
        __syncthreads();
        // regroup the chunk: hits from the front of s_list, misses from its back, so that the waves below shade either
        // hits or misses (one mixed wave per chunk at most) instead of running both code paths with half their lanes
#pragma unroll 1
        for (uint32_t kk = 0; kk < RP_CHUNK / 256; ++kk) {
            const uint32_t i = chunk * RP_CHUNK + kk * 256 + threadIdx.x;
            const bool valid = i < n;
            uint32_t pp = 0;
            bool is_hit = false;
            if (valid) {
                pp = (FIRST && !order) ? base + i : order[i];
                is_hit = (STREAM ? __float_as_int(rp_ld2<true>(reinterpret_cast<const float2 *>(ps.hit_ids), pp).x) : ps.hit_ids[pp].x) >= 0;
            }
            const uint32_t ah = rp_wave_append(&s_nhit, valid && is_hit);
            if (valid && is_hit) s_list[ah] = pp;
            const uint32_t am = rp_wave_append(&s_nmiss, valid && !is_hit);
            if (valid && !is_hit) s_list[RP_CHUNK - 1 - am] = pp;
        }
        __syncthreads();
        const uint32_t chunk_hits = s_nhit, chunk_n = s_nhit + s_nmiss;
        // north_star's "regroup rays by material before the BSDF stages", fused into this compaction (no launch, no extra pass over the
        // queue): the hits of the chunk are ordered by material id -- a counting sort through 64 LDS bins, s_next as the second buffer (it is
        // empty here). The order inside a bin depends on the LDS atomics, a path's result does not depend on its position. Measured
        // (profiles/r03_notes.md): it does not pay -- every material runs the same BSDF code, only its parameters differ -- so it stays an
        // experiment behind RPTR_REGROUP=1.
        if (f.regroup_materials && chunk_hits > 64u) {
            __shared__ uint32_t s_bin[64];
            __shared__ unsigned char s_key[RP_CHUNK];
            if (threadIdx.x < 64) s_bin[threadIdx.x] = 0;
            __syncthreads();
            for (uint32_t il = threadIdx.x; il < chunk_hits; il += 256) {
                const uint32_t pp = s_list[il];
                const int2 ids = ps.hit_ids[pp];
                const int prim = __float_as_int(ps.hit_tuv[pp].w);
                const RpGeomRecord &g = sc.geoms[sc.insts[ids.x].geometry_base + ids.y];
                const uint32_t key = (uint32_t)rp_hit_material_id(g, (uint32_t)prim) & 63u;
                s_key[il] = (unsigned char)key;
                atomicAdd(&s_bin[key], 1u);
            }
            __syncthreads();
            if (threadIdx.x < 64) { // exclusive prefix over the 64 bins (one wave)
                uint32_t v = s_bin[threadIdx.x], incl = v;
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t o = __shfl_up(incl, off);
                    if ((int)threadIdx.x >= off) incl += o;
                }
                s_bin[threadIdx.x] = incl - v;
            }
            __syncthreads();
            for (uint32_t il = threadIdx.x; il < chunk_hits; il += 256) s_next[atomicAdd(&s_bin[s_key[il]], 1u)] = s_list[il];
            __syncthreads();
            for (uint32_t il = threadIdx.x; il < chunk_hits; il += 256) s_list[il] = s_next[il];
            __syncthreads();
        }
#pragma unroll 1
        for (uint32_t kk = 0; kk < RP_CHUNK / 256; ++kk) {
            RP_RELOAD_ARGS
            const uint32_t il = kk * 256 + threadIdx.x; // position in the regrouped chunk
            bool alive = false;      // path continues with a new ray
            bool has_shadow = false; // a shadow query is issued
            uint32_t p = 0;
            // The body runs in three stretches with the whole wave converged in between, so that the binned-RIS
            // candidates of all tri-light samples of the wave can be spread over all 64 lanes (A: up to the choice of
            // the light kind; cooperative candidate evaluation; B: next-event estimation; C: BSDF sample + state).
            bool hit_lane = false;  // this lane shades a hit
            bool nee = false;       // ... and samples direct light
            bool nee_tri = false;   // ... from the triangle lights (needs the bin's contributions)
            bool terminate = false;
            RpRng rng;
            rng.s = rng.index = rng.pix = 0u;
            float total_t = 0.f, prev_bounce_pdf = 0.f, geometry_scale = 0.f;
            V3 ray_origin = v3s(0.f), ray_dir = v3s(0.f), throughput = v3s(0.f), illum = v3s(0.f), scatter_throughput = v3s(0.f);
            V3 ip_p = v3s(0.f), gn = v3s(0.f), nn = v3s(0.f), w_o = v3s(0.f), v_x = v3s(0.f), v_y = v3s(0.f);
            int bounce = 0;
            int aov_px = -1; // FIRST: local pixel whose AOVs this path writes
            V2 aov_jitter = v2(0.0f, 0.0f);
            RpMaterial mat;
            M2 tex_fp{v2(0.f, 0.f), v2(0.f, 0.f)}; // TEX: texture_footprint (pt_megakernel.glsl:336-352)
            V2 dir_sample = v2(0.f, 0.f), sel_sample = v2(0.f, 0.f);
            RpLightBin bin;
            bin.bin_begin = bin.bin_end = 0;
            bin.sel_p = 0.f;
            const int output_channel = f.rp.output_channel;
            const float sun_w = f.sp.sun_radiance[3];
            // ---------------- A
            bool present = il < chunk_n; // this lane shades a path
            int first_lx = 0, first_ly = 0;
            uint32_t first_sslot = 0;
            V3 first_du_dv[2] = {v3s(0.f), v3s(0.f)}; // FIRST && TEX: the image-plane axes of the path's camera
            if (present) {
                p = il < chunk_hits ? s_list[il] : s_list[RP_CHUNK - 1 - (il - chunk_hits)];
                // bounce 0: the camera ray again; ids of tile padding name no pixel sample (the first queue is the identity)
                if (FIRST) present = rp_primary_ray_ex<TABLE>(f, p, rng, ray_dir, first_lx, first_ly, first_sslot, ray_origin, TEX ? first_du_dv : nullptr);
            }
            if (present) {
                my_closest++;
                if (FIRST) { // init_shading_sample_state (shading_interface.glsl:20-22)
                    if (f.alpha_test && (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM))
                        rng.s = rp_ld1<SH>(reinterpret_cast<const uint32_t *>(ps.ray_d), 4u * p + 3u); // alpha tests of the first extend may have drawn from it
                    if (f.aov_albedo_roughness) { // the first sample of the (last) frame (of the batch) writes the AOVs
                        const RpSlotFrame sf = rp_slot_frame(f, first_sslot);
                        if (sf.sample_index == sf.frame_id && int(sf.frame) == f.batch_frames - 1) {
                            aov_px = first_ly * f.width + first_lx;
                            if (TABLE) aov_jitter = rp_screen_jitter(f, sf.frame_offset, sf.frame_id);
                        }
                    }
                    throughput = v3s(1.0f);
                    illum = v3s(0.0f);
                    prev_bounce_pdf = 2.e16f;
                    total_t = 0.0f;
                    bounce = 0;
                    if (TEX) { // :341-351
                        const V3 dpdx = (first_du_dv[0] / float(f.width)) * f.rp.pixel_radius, dpdy = (first_du_dv[1] / float(f.height)) * f.rp.pixel_radius;
                        tex_fp = rp_dpdxy_to_footprint(ray_dir, dpdx, dpdy);
                    }
                } else {
                    const float4 ro4 = rp_ld4<SH>(ps.ray_o, p), rd4 = rp_ld4<SH>(ps.ray_d, p);
                    const float4 thr4 = rp_ld4<SH>(ps.thr, p);
                    const float4 il4 = rp_ld4<SH>(ps.illum, p);
                    rng = rp_rng_resume<TABLE>(f, p, __float_as_uint(rd4.w));
                    total_t = ro4.w;
                    ray_origin = xyz(ro4);
                    ray_dir = xyz(rd4);
                    throughput = xyz(thr4);
                    illum = xyz(il4);
                    prev_bounce_pdf = thr4.w;
                    bounce = __float_as_int(il4.w);
                    if (TEX) {
                        const float4 fp = rp_ld4<SH>(ps.footprint, p);
                        tex_fp = M2{v2(fp.x, fp.y), v2(fp.z, fp.w)};
                    }
                }
                const float4 hit4 = rp_ld4<STREAM>(ps.hit_tuv, p);
                int2 ids;
                if (STREAM) {
                    const float2 idf = rp_ld2<true>(reinterpret_cast<const float2 *>(ps.hit_ids), p);
                    ids = make_int2(__float_as_int(idf.x), __float_as_int(idf.y));
                } else
                    ids = ps.hit_ids[p];
                if (ids.x < 0) {
                    // miss: pt_megakernel.glsl:480-489
                    illum = illum + throughput * rp_compute_sky_illum(f, ray_dir, prev_bounce_pdf);
                    rp_st4<SH>(ps.illum, p, f4(illum, __int_as_float(bounce)));
                    if (FIRST && aov_px >= 0) { // pt_megakernel.glsl:482-487
                        rp_store_geometry_aovs(f, aov_px, v3s(0.0f), v3s(2.e32f), aov_jitter);
                        rp_store_material_aovs(f, aov_px, v3s(0.0f), 1.0f, 1.0f);
                    }
                } else {
                    hit_lane = true;
                    my_hits++;
                    // ---- hit attributes, pt_megakernel.glsl:495-572
                    const float4 *ip = reinterpret_cast<const float4 *>(sc.insts + ids.x);
                    const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
                    const int4 meta = *reinterpret_cast<const int4 *>(ip + 3);
                    const RpGeomRecord g = sc.geoms[meta.y + ids.y];
                    const uint32_t prim = uint32_t(__float_as_int(hit4.w));
                    // transpose(mat3(world_to_object)): its columns are the rows of world_to_object
                    const M3 n2w{v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z)};
                    RpHit hit = rp_calc_hit_attributes(g, hit4.x, prim, hit4.y, hit4.z, n2w);
                    // :578-580
                    float approx_tri_solid_angle = len3(hit.geo_normal);
                    hit.geo_normal = hit.geo_normal / approx_tri_solid_angle;
                    approx_tri_solid_angle *= fabsf(dot3(hit.geo_normal, ray_dir)) / (hit.dist * hit.dist);
                    // :582-606
                    total_t += hit.dist;
                    const RpTexCoord tc = TEX ? rp_hit_texcoord(hit.uv, tex_fp, ray_dir, hit.geo_normal, hit.tangent, hit.bitangent_l, total_t) : rp_texcoord(hit.uv);
                    geometry_scale = total_t;
                    w_o = -ray_dir;
                    ip_p = ray_origin + hit.dist * ray_dir;
                    gn = hit.geo_normal;
                    nn = hit.normal;
                    const RptrBaseMaterial mp = sc.materials[hit.material_id];
                    // :624-633
                    if (dot3(w_o, gn) < 0.0f) {
                        if ((mp.flags & RPTR_BASE_MATERIAL_VOLUME) != 0) {
                            ip_p = ray_origin;
                            hit.dist = 0.0f;
                        } else if ((mp.flags & RPTR_BASE_MATERIAL_ONESIDED) == 0) {
                            nn = -nn;
                            gn = -gn;
                        }
                    }
                    // :634-654 normal mapping
                    if (TEX && mp.normal_map != -1) {
                        V3 t_y = norm3(cross3(hit.normal, hit.tangent));
                        V3 t_x = cross3(t_y, hit.normal);
                        t_x = t_x * len3(hit.tangent);
                        t_y = t_y * hit.bitangent_l;
                        const float4 tx = rp_texture_lod(sc, mp.normal_map, hit.uv, float(bounce)); // :642-648
                        V3 map_nrm = v3(2.0f * tx.x - 1.0f, 2.0f * tx.y - 1.0f, 1.0f * tx.z - 0.0f);
                        map_nrm.z = sqrtf(fmaxf(1.0f - map_nrm.x * map_nrm.x - map_nrm.y * map_nrm.y, 0.0f));
                        const V3 t_z = f.sp.normal_z_scale * nn;
                        nn = norm3((t_x * map_nrm.x + t_y * map_nrm.y) + t_z * map_nrm.z);
                    }
                    // :656-668
                    {
                        const float nw = dot3(w_o, nn);
                        const float gnw = dot3(w_o, gn);
                        if (nw * gnw <= 0.0f) {
                            const float blend = gnw / (gnw - nw);
                            nn = norm3(mix3(gn, nn, blend - RP_EPSILON));
                        }
                    }
                    // :677-678
                    v_y = norm3(cross3(nn, hit.tangent));
                    v_x = cross3(v_y, nn);

                    // ---- shade_base_material, rendering/mc/shade_base_material.glsl:14-96
                    V3 emit;
                    rp_unpack_material<VARIANT, TEX>(sc, mat, emit, mp, tc);
                    scatter_throughput = throughput;
                    if (FIRST && aov_px >= 0) { // pt_megakernel.glsl:670-673, shade_base_material.glsl:28-31
                        rp_store_geometry_aovs(f, aov_px, nn, ip_p, aov_jitter);
                        rp_store_material_aovs(f, aov_px, throughput * mat.base_color, mat.roughness, mat.ior);
                    }
                    if (output_channel == 0 && !eq3(emit, v3s(0.0f))) {
                        // wpdf_direct_light, nee_interface.glsl:52-61 + lights_linear.glsl:129-137
                        const float light_pdf = (1.0f - f.sp.sun_radiance[3]) * (1.0f / (float(f.num_bins) * approx_tri_solid_angle));
                        const float w = rp_nee_mis(prev_bounce_pdf, light_pdf);
                        illum = illum + w * scatter_throughput * emit;
                    }
                    if (output_channel != 0) {
                        const float reliability = powf(0.25f, float(bounce));
                        if (output_channel == 1)
                            illum = illum + scatter_throughput * mat.base_color * reliability;
                        else if (output_channel == 2)
                            illum = illum + nn * reliability;
                        else if (output_channel == 3)
                            illum = illum + ip_p * reliability;
                    }
                    terminate = (bounce + 1 >= f.rp.max_path_depth);
                    if (!terminate && output_channel == 0) {
                        // ---- sample_direct_light, rendering/mc/nee.glsl:32-90: the random numbers and the kind of light
                        nee = true;
                        dir_sample = rp_draw2<TABLE>(f, rng, rp_bounce_dim(bounce) + 2u); // DIM_POSITION_X
                        sel_sample = rp_draw2<TABLE>(f, rng, rp_bounce_dim(bounce));      // DIM_LIGHT_SEL_1
                        if (LIGHTS && !(sel_sample.x <= sun_w)) {
                            nee_tri = true;
                            sel_sample.x = (sel_sample.x - sun_w) / (1.0f - sun_w);
                            bin = rp_choose_light_bin(sc, f, sel_sample.x);
                        }
                    }
                }
            }
            // ---------------- the bin contributions of every tri-light sample of this wave, 64 candidates at a time
            const float *my_contrib = nullptr;
            if (LIGHTS) {
                const unsigned long long want = __ballot(nee_tri);
                if (want != 0ull) {
                    const uint32_t lane = rp_lane_id();
                    float *wreq = s_ris_req + (threadIdx.x >> 6) * (64 * 8);
                    float *wcon = s_ris_contrib + (threadIdx.x >> 6) * (64 * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE);
                    const uint32_t rank = (uint32_t)__popcll(want & ((1ull << lane) - 1ull));
                    const uint32_t nreq = (uint32_t)__popcll(want);
                    if (nee_tri) {
                        float *q = wreq + rank * 8;
                        q[0] = ip_p.x;
                        q[1] = ip_p.y;
                        q[2] = ip_p.z;
                        q[3] = nn.x;
                        q[4] = nn.y;
                        q[5] = nn.z;
                        q[6] = __int_as_float(bin.bin_begin);
                        q[7] = __int_as_float(bin.bin_end);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const uint32_t jobs = nreq * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE;
                    for (uint32_t j = lane; j < jobs; j += 64u) {
                        const uint32_t qi = j / RPTR_BINNED_LIGHTS_BIN_MAX_SIZE, ci = j % RPTR_BINNED_LIGHTS_BIN_MAX_SIZE;
                        const float *q = wreq + qi * 8;
                        const V3 hp = v3(q[0], q[1], q[2]), hn = v3(q[3], q[4], q[5]);
                        const int bb = __float_as_int(q[6]), be = __float_as_int(q[7]);
                        wcon[j] = rp_tri_light_contribution(sc, bb + (int)ci, be, hp, hn);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    my_contrib = wcon + rank * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE;
                }
            }
            // ---------------- B: next-event estimation
            if (nee) {
                RP_RELOAD_ARGS
                V3 nee_l = v3s(0.0f);
                V3 light_dir = v3s(0.0f);
                float light_dist = 2.e16f, light_pdf = 0.0f, mis_pdf = 0.0f;
                if (!nee_tri) {
                    sel_sample.x /= sun_w;
                    light_dir = rp_sample_sun_dir(ld3(f.sp.sun_dir), f.sp.sun_cos_angle, dir_sample);
                    light_pdf = rp_sun_dir_pdf(f.sp.sun_cos_angle);
                    nee_l = nee_l + (v3s(1.0f) / v3s(light_pdf)) * (ld3(f.sp.sun_radiance) / sun_w);
                    light_pdf *= sun_w;
                    mis_pdf = light_pdf;
                } else {
                    float tri_mis_wpdf = 0.0f;
                    nee_l = nee_l + rp_finish_tri_light_sample(sc, f, bin, [&](int ci) { return my_contrib[ci]; }, ip_p, dir_sample, sel_sample.y, light_dir,
                                                               light_dist, light_pdf, tri_mis_wpdf) /
                                        (1.0f - sun_w);
                    light_pdf *= 1.0f - sun_w;
                    if (mis_pdf == 0.0f) mis_pdf = tri_mis_wpdf * (1.0f - sun_w);
                }
                if (light_pdf > 0.0f && dot3(light_dir, gn) * dot3(light_dir, nn) > 0.0f) {
                    // raytrace_test_visibility is deferred to the connect stage; everything that
                    // does not depend on its answer is evaluated here (nee.glsl:73-84)
                    const float bsdf_pdf = rp_eval_bsdf_wpdf<VARIANT>(mat, nn, w_o, light_dir);
                    const float epsilon = rp_geometry_scale_to_tmin(ip_p, geometry_scale);
                    const bool needs_ray = (light_dist - 2.f * epsilon > 0.0f); // pt_megakernel.glsl:222-227
                    if (needs_ray) my_shadow++; // the reference traces it even when bsdf_pdf < 0
                    if (bsdf_pdf >= 0.0f) {
                        const V3 bsdf = rp_eval_bsdf<VARIANT>(mat, nn, w_o, light_dir);
                        const float w = rp_nee_mis(mis_pdf, bsdf_pdf);
                        nee_l = nee_l * ((w * fabsf(dot3(light_dir, nn))) * bsdf);
                        const V3 c = scatter_throughput * nee_l;
                        if (needs_ray) {
                            has_shadow = true; // (its origin and offset: ray_o below -- ip_p and the path length the offset was computed from)
                            sq.d[p] = f4(light_dir, light_dist - epsilon);
                            sq.contrib[p] = f4(c, 0.0f);
                        } else
                            illum = illum + c; // visibility defaults to true
                    }
                }
            }
            // ---------------- C: continuation
            if (hit_lane) {
                RP_RELOAD_ARGS
                if (!terminate && f.rp.glossy_only_mode != 0 && !(mat.roughness < 0.1f && mat.ior != 1.0f)) terminate = true;
                if (!terminate) {
                    const uint32_t vertex_dim = rp_bounce_dim(bounce) + 4u; // behind RANDOM_SHIFT_DIM(rng, DIM_LIGHT_END)
                    const V2 lobe_sample = rp_draw2<TABLE>(f, rng, vertex_dim + 2u); // DIM_LOBE
                    const V2 dir_sample2 = rp_draw2<TABLE>(f, rng, vertex_dim);      // DIM_DIRECTION_X
                    V3 w_i = v3s(0.0f);
                    float sampling_pdf = 0.0f, mis_pdf = 0.0f;
                    V3 bsdf;
                    if (VARIANT == RPTR_VARIANT_SIMPLE)
                        bsdf = rp_sample_simple_brdf(mat, nn, w_i, sampling_pdf, mis_pdf, dir_sample2);
                    else if (VARIANT == RPTR_VARIANT_GLTF_TRANSMISSION)
                        bsdf = rp_sample_gltf_t_brdf(mat, nn, w_o, w_i, sampling_pdf, mis_pdf, dir_sample2, lobe_sample, v_x, v_y);
                    else
                        bsdf = rp_sample_gltf_brdf(mat, nn, w_o, w_i, sampling_pdf, mis_pdf, dir_sample2, lobe_sample, v_x, v_y);
                    ++bounce;
                    if (eq3(bsdf, v3s(0.f)) || mis_pdf == 0.f || !(dot3(w_i, nn) * dot3(w_i, gn) > 0.0f))
                        terminate = true;
                    else {
                        throughput = throughput * bsdf;
                        prev_bounce_pdf = mis_pdf;
                        // pt_megakernel.glsl:698-709
                        if (TEX && dot3(w_i, nn) * dot3(w_o, nn) > -0.999f) tex_fp = rp_reflect_footprint(w_i, ray_dir, tex_fp);
                        ray_dir = w_i;
                        ray_origin = ip_p;
                        // :713-730 Russian roulette
                        bool survive = true;
                        if (bounce >= f.rp.rr_path_depth) {
                            const float prefix_weight = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
                            float rr_prob = prefix_weight;
                            const float rr_sample = rp_draw1<TABLE>(f, rng, vertex_dim + 3u); // DIM_RR: the unused free-path slot of this vertex
                            rr_prob = (bounce > 6) ? fminf(0.95f, rr_prob) : fminf(1.0f, rr_prob);
                            if (rr_sample < rr_prob)
                                throughput = throughput / rr_prob;
                            else
                                survive = false;
                        }
                        if (survive) {
                            alive = true;
                            rp_st4<SH>(ps.ray_d, p, f4(ray_dir, __uint_as_float(rng.s)));
                            rp_st4<SH>(ps.thr, p, f4(throughput, prev_bounce_pdf));
                            if (TEX) rp_st4<SH>(ps.footprint, p, make_float4(tex_fp.c0.x, tex_fp.c0.y, tex_fp.c1.x, tex_fp.c1.y));
                        }
                    }
                }
                // the vertex (ip_p: a surviving path's ray_origin is ip_p) and the path length up to it: what the shadow ray and the continuation
                // ray start from (their offset = rp_geometry_scale_to_tmin of the two, recomputed by connect / extend)
                if (alive || has_shadow) rp_st4<SH>(ps.ray_o, p, f4(ip_p, total_t));
                rp_st4<SH>(ps.illum, p, f4(illum, __int_as_float(bounce)));
            }
            if (STREAM) {
                const bool any = alive || has_shadow;
                const uint32_t at = rp_wave_append(&s_nn, any);
                if (any) s_next[at] = p | (alive ? RP_ITEM_CONT : 0u) | (has_shadow ? RP_ITEM_SHADOW : 0u);
            } else {
                const uint32_t at = rp_wave_append(&s_nn, alive);
                if (alive) s_next[at] = p;
                const uint32_t sat = rp_wave_append(&s_ns, has_shadow);
                if (has_shadow) s_shadow[sat] = p;
            }
        }
        __syncthreads();
        if (!LOCAL) {
            rp_block_flush(s_next, s_nn, next_queue, next_count, &s_base);
            __syncthreads();
            rp_block_flush(s_shadow, s_ns, sq.ids, shadow_count, &s_base);
            __syncthreads();
        }
    }
    my_closest = rp_wave_sum_u32(my_closest);
    my_shadow = rp_wave_sum_u32(my_shadow);
    my_hits = rp_wave_sum_u32(my_hits);
    if (rp_lane_id() == 0) {
        atomicAdd(&s_stat[0], my_closest);
        atomicAdd(&s_stat[1], my_shadow);
        atomicAdd(&s_stat[2], my_hits);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_stat[0]) atomicAdd(&ctr->rays_closest, (unsigned long long)s_stat[0]);
        if (s_stat[1]) atomicAdd(&ctr->rays_shadow, (unsigned long long)s_stat[1]);
        if (s_stat[2]) atomicAdd(&ctr->hits_shaded, (unsigned long long)s_stat[2]);
    }
    if (LOCAL) {
        local_next = s_next;
        local_shadow = s_shadow;
        n_next = s_nn;
        n_shadow = s_ns;
    }
}
// The Lambert instantiation without lights, textures or tables (C2, C4, C5) needs 97-105 VGPRs: just above the 96 that let a SIMD hold
// five waves instead of four. Its later-bounce launches wait on dependent loads 70-80 % of their wave-cycles (hit -> instance -> geometry
// -> vertices -> material; profiles/pmc_traffic.json wait_any_frac), so the fifth wave pays: compiled for five the two kernels fit 96
// VGPRs without scratch; A/B on one box (tools/ab.sh, build variants): shade launches 0.439 -> 0.425 ms per C2 frame, the pipelined
// frame 1.32 -> 1.30 ms, C5 3.20 -> 3.15, C4 unchanged. -DRP_SHADE_WAVES_LEAN=4: the old bound.
// (After the build lost the SLP vectoriser the two kernels need 87 / 96 VGPRs; compiled for SIX waves they fit 80 with 0 / 8 bytes of scratch.
// Their exclusive times do not move, the pipelined frames do -- fewer registers per block leave room for the other frames' waves: C2 1.216 ->
// 1.179 ms (three pairs of runs), C4 4.74 -> 4.65, C5 2.96 -> 2.91.)
// (Seven: 72 VGPRs, 12 / 8 bytes of scratch -- and now the exclusive launches gain too, shade 0.421 -> 0.397 ms per C2 frame: the later bounces
// wait on dependent loads and a seventh wave hides more of them; pipelined C2 1.150 / 1.136 / 1.164 -> 1.114 / 1.131 / 1.129, C4 4.47 -> 4.42.)
#ifndef RP_SHADE_WAVES_LEAN
#define RP_SHADE_WAVES_LEAN 7
#endif
template <int VARIANT, bool LIGHTS, bool TEX, bool TABLE>
constexpr int rp_shade_waves() { return (VARIANT == RPTR_VARIANT_SIMPLE && !LIGHTS && !TEX && !TABLE) ? RP_SHADE_WAVES_LEAN : RP_SHADE_WAVES; }
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX, bool TABLE>
__global__ __launch_bounds__(256, (rp_shade_waves<VARIANT, LIGHTS, TEX, TABLE>())) void rp_k_shade(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, const uint32_t *order,
                                                  const uint32_t *count_ptr, uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count,
                                                  RpCounters *ctr) {
    uint32_t *ln = nullptr, *ls = nullptr;
    uint32_t nn = 0, ns = 0;
    rp_shade_body<VARIANT, FIRST, LIGHTS, TEX, false, TABLE>(sc, f, ps, sq, order, *count_ptr, next_queue, next_count, shadow_count, ctr, ln, nn, ls, ns);
}

// ------------------------------------------------------------------ tail: the late bounces of a frame in ONE launch
// From some bounce on a frame's queues hold a few thousand paths, and what a bounce then costs is its three launches
// (a command-processor packet each, ~14 us when several frames are in flight: profiles/r01_notes.md), not its rays. Paths are
// independent, so the rest of the frame needs no grid-wide step: a block takes RP_TAIL_CHUNK paths of the bounce's queue and runs
// them to the end -- extend, shade, connect per bounce on block-local lists in LDS, the same device code as the stand-alone
// kernels (results are bit-identical, tests/test_gpu_parity.py) -- before it takes the next chunk.
#define RP_TAIL_CHUNK 256
template <int VARIANT, bool LIGHTS, bool TEX, bool ALPHA, bool SINGLE, bool TABLE>
__global__ __launch_bounds__(256, 1) void rp_k_tail(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, const uint32_t *queue, RpCounters *ctr,
                                                    int first_bounce, int *gstack) {
    // One arena for the phases that take turns (round 4, as in rp_k_frame): the LDS stacks of the closest-hit traversal, the shade phase's
    // scratch (regrouped list, light-candidate exchange) and the LDS stacks of the shadow-ray traversal -- 33 KB per block instead of 64 (37
    // instead of 88 with triangle lights). A tail block issues next to nothing for 0.16 ms of every frame; with eleven frames in flight one
    // or two of them sit on every CU at any time, and what they hold is LDS the traversal blocks of the other frames want.
    constexpr uint32_t STACK_WORDS = RP_LDS_STACK * RP_TRAVERSE_BLOCK;
    constexpr uint32_t SCRATCH_WORDS = RP_CHUNK + (LIGHTS ? RP_SHADE_RIS_REQ_FLOATS + RP_SHADE_RIS_CONTRIB_FLOATS : 0);
    __shared__ __attribute__((aligned(16))) uint32_t s_arena[STACK_WORDS > SCRATCH_WORDS ? STACK_WORDS : SCRATCH_WORDS];
    __shared__ uint32_t s_next[RP_CHUNK], s_shadow[RP_CHUNK];
    __shared__ uint32_t s_cur[RP_TAIL_CHUNK];
    __shared__ uint32_t s_cursor[2];
    int *const stack = reinterpret_cast<int *>(s_arena);
    RpShadeLds lds;
    lds.next = s_next;
    lds.shadow = s_shadow;
    lds.list = s_arena;
    lds.ris_req = reinterpret_cast<float *>(s_arena + RP_CHUNK);
    lds.ris_contrib = reinterpret_cast<float *>(s_arena + RP_CHUNK + RP_SHADE_RIS_REQ_FLOATS);
    const uint32_t n_total = ctr->bounce[first_bounce].queue_count;
    const uint32_t nchunks = (n_total + RP_TAIL_CHUNK - 1) / RP_TAIL_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        uint32_t n = min((uint32_t)RP_TAIL_CHUNK, n_total - chunk * RP_TAIL_CHUNK);
        __syncthreads(); // the previous chunk is done with s_cur
        if (threadIdx.x < n) s_cur[threadIdx.x] = queue[chunk * RP_TAIL_CHUNK + threadIdx.x];
        for (int b = first_bounce; b < f.rp.max_path_depth && n > 0; ++b) {
            if (threadIdx.x < 2) s_cursor[threadIdx.x] = 0;
            __syncthreads();
            rp_extend_body<false, false, ALPHA, SINGLE, true, TABLE, 0, false, true>(sc, f, ps, s_cur, n, &s_cursor[0], ctr, gstack, 0u, stack);
            __syncthreads();
            uint32_t *next = nullptr, *shadow = nullptr;
            uint32_t n_next = 0, n_shadow = 0;
            rp_shade_body<VARIANT, false, LIGHTS, TEX, true, TABLE, false, true>(sc, f, ps, sq, s_cur, n, nullptr, nullptr, nullptr, ctr, next, n_next, shadow, n_shadow, 0u,
                                                                                  &lds);
            __syncthreads();
            rp_connect_body<false, ALPHA, SINGLE, true, 0, false, true>(sc, f, ps, sq, shadow, n_shadow, &s_cursor[1], ctr, gstack, stack);
            __syncthreads();
            if (threadIdx.x < n_next) s_cur[threadIdx.x] = next[threadIdx.x]; // survivors: at most n <= RP_TAIL_CHUNK
            n = n_next;
        }
    }
}

// ------------------------------------------------------------------ frame: the whole frame in ONE launch, driven from device-side queues
// (the reference's megakernel is one dispatch per frame: vulkan/render_pipeline_vulkan.cpp:253-261, render_vulkan.cpp:2961-3059)
//
// The stand-alone stages make a frame a chain of ~10 dependent launches, and every launch lasts as long as its slowest ray: ONE frame
// rendered alone takes 2.06 ms where frames in flight reach 1.3 ms (C2), and a 1/8 frame (one rank of an 8-GPU split) 0.79 ms instead of
// 0.18 (profiles/r03_notes.md section 4). Paths are independent, so nothing in a frame needs a grid-wide step: rp_k_frame replaces the
// launch boundaries by CHUNK-level dependencies. A block
//   1. takes a work item: a SLOT of RP_CHUNK consecutive entries of some bounce's queue -- a slot of a bounce >= 1 from the ring of
//      ready slots if there is one (deeper bounces first: paths end sooner, queues stay short), else up to k0 slots of bounce 0 (the
//      identity over the path ids; several at a time so that the block-local traversal has rays to refill its lanes with),
//   2. runs the slot's paths through extend -> shade -> connect with the device code of the stand-alone kernels (rp_*_body, LOCAL lists in
//      LDS: what rp_k_tail does), so the image is bit-identical to theirs,
//   3. appends the survivors to the NEXT bounce's global queue -- whichever block gets the slot they land in continues them, which is what
//      keeps 64 lanes per wave busy on the second and third bounce where rp_k_tail's block-local lists would thin out -- or, from bounce
//      n_pub - 1 on (queues of a few thousand paths), keeps them and runs them to their end like rp_k_tail.
// A straggler -- a ray that grazes the height field through hundreds of nodes -- now holds back its own block, not the frame.
//
// Queue protocol. Nobody polls a slot: the events that make work say so. (A first version had every idle block walk tail / head / done
// words and compare-and-swap a shared head: 5 M claim attempts for 11 thousand slots, 400 ms per frame -- one device word sustains ~88
// atomics per microsecond, MI355X_MICROARCH.md "dequeue".)
//   tail[b]       entries reserved in bounce b's queue: one atomicAdd per chunk of survivors (bounce 0: preset to the number of paths)
//   commit[b][s]  entries of slot s whose ids are in memory. The producer whose add makes it RP_CHUNK pushes the slot into the ring.
//   done[b]       slots of bounce b whose paths have been traced, shaded and whose survivors have been published
//   final[b]      bounce b's tail will not grow: bounce b - 1 is final and done[b - 1] covers all of its slots (bounce 0: preset). Whoever
//                 observes that first (after its own add to done[b - 1], or after setting final[b - 1]) sets the flag -- a compare-and-swap
//                 elects one -- and pushes the bounce's last, partly filled slot. The frame is complete when the last published bounce is
//                 final and done.
//   ring          64-bit entries (frame epoch | bounce | slot | entries; no memset of the ring), written at a position reserved with an
//                 atomicAdd on ring_tail, read by the block that holds the position's TICKET (an atomicAdd on ring_head: rp_fq_claim)
//   head0         tickets of bounce 0 (atomicAdd: never fails)
// Visibility (MI355X_MICROARCH.md "inter-workgroup visibility"): per-XCD L2s are not coherent and a CU's L1 is never refreshed by other
// CUs' stores. Consumers read path state and ids around the L1 (SH accessors above); a producer drains its stores in every wave, joins at
// a barrier, and ONE lane issues an agent-scope release (buffer_wbl2 sc1 + s_waitcnt) before the commit -- measured: sc1 "write-through"
// stores drained with s_waitcnt vmcnt(0) alone were NOT enough (stale ids on the second frame of a handle, faults at 4 blocks per CU;
// with the release: bit-identical on every size tried) --; the control words are agent-scope atomics.
#define RP_FQ_MAX 8 // bounces whose queues can be global; later bounces always run block-local
#ifndef RP_FRAME_WAVES
#define RP_FRAME_WAVES 4 // blocks per CU the frame kernel is compiled for (VGPR budget: its shade phase)
#endif
struct RpFqState {
    uint32_t tail[RP_FQ_MAX], done[RP_FQ_MAX], final[RP_FQ_MAX];
    uint32_t head0;                 // slots of bounce 0 handed out
    uint32_t ring_tail, ring_head;  // the ring of ready slots
    uint32_t complete;              // the frame is done: idle blocks leave
    uint32_t polls, idle_polls;     // diagnostics: claim attempts, attempts that found nothing to do (added once per block)
    uint32_t timeout;               // a block gave up waiting for work that never came (a protocol error: the host reports it)
    uint32_t bad_ids;               // queue entries that named no path (>= capacity): never on a correct run, reported by the host
    unsigned long long t_claim, t_extend, t_shade, t_connect, t_publish, t_fence, t_total; // -DRP_FRAME_PROF: 100 MHz ticks summed over the blocks (lane 0)
};
struct RpFrameQueues {
    RpFqState *st;
    uint32_t *ids;               // path ids of bounce b >= 1 at ids + (b - 1) * capacity
    uint32_t *commit;            // one word per slot, bounce b >= 1 at commit + (b - 1) * slots
    unsigned long long *ring;    // ready slots (RP_FQ_MAX * slots entries: every slot is pushed once per frame at most)
    uint32_t slots;
    uint32_t epoch;              // this frame's tag in ring entries (1 .. 65535; the ring is zeroed when it is allocated)
    int32_t n_pub;               // bounces 0 .. n_pub - 1 have global queues (1 <= n_pub <= RP_FQ_MAX)
    int32_t k0;                  // slots of bounce 0 a block takes at a time (1 when n_pub == 1)
    uint32_t capacity;           // path ids are below this
    uint32_t dbg;                // RPTR_FRAME_DBG (experiments): bit 0 no release fence before a commit
};
struct RpFqWork {
    uint32_t b, slot, m, n; // bounce (~0u: the frame is complete), first slot, slots, entries
};
RP_DEV uint32_t rp_fq_ld(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV uint32_t rp_fq_add(uint32_t *p, uint32_t v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
RP_DEV uint32_t *rp_fq_ids(const RpFrameQueues &fq, uint32_t b) { return fq.ids + size_t(b - 1u) * fq.capacity; }
RP_DEV uint32_t *rp_fq_commit(const RpFrameQueues &fq, uint32_t b) { return fq.commit + size_t(b - 1u) * fq.slots; }
// every wave, before a block barrier behind which ANOTHER wave reads what this one stored device-coherently: a barrier orders the waves,
// not their stores (sc1 accesses go around the L1 that orders the plain ones of a workgroup)
RP_DEV void rp_drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// (one lane) a slot of bounce b with n entries is ready
RP_DEV void rp_fq_push(const RpFrameQueues &fq, uint32_t b, uint32_t slot, uint32_t n) {
    const uint32_t i = rp_fq_add(&fq.st->ring_tail, 1u);
    const unsigned long long e = ((unsigned long long)fq.epoch << 48) | ((unsigned long long)b << 44) | ((unsigned long long)slot << 12) | n;
    __hip_atomic_store(fq.ring + i, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (one lane) after an add to done[b1 - 1] or after final[b1 - 1] was set: bounce b1 may have become final (b1 == n_pub: the frame complete)
RP_DEV void rp_fq_try_finalize(const RpFrameQueues &fq, uint32_t b1) {
    RpFqState *const st = fq.st;
    for (;; ++b1) {
        const uint32_t b = b1 - 1u;
        if (rp_fq_ld(&st->final[b]) == 0u) return;
        rp_drain_stores(); // (a final bounce's tail stands still: read it after the flag)
        const uint32_t t = rp_fq_ld(&st->tail[b]), dn = rp_fq_ld(&st->done[b]);
        if (dn != (t + RP_CHUNK - 1u) / RP_CHUNK) return;
        if (b1 == (uint32_t)fq.n_pub) {
            __hip_atomic_store(&st->complete, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        uint32_t expect = 0u;
        if (!__hip_atomic_compare_exchange_strong(&st->final[b1], &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        // this lane made bounce b1 final: its last slot, if partly filled, will never be completed by a producer
        const uint32_t t1 = rp_fq_ld(&st->tail[b1]);
        if (t1 % RP_CHUNK != 0u) rp_fq_push(fq, b1, t1 / RP_CHUNK, t1 % RP_CHUNK);
    }
}
// (one lane) the block's next work item. Every block holds a TICKET of the ring -- position `ticket` belongs to it and to nobody else, so
// taking a ready slot is a load of the block's own entry, no compare-and-swap race of a thousand blocks for one head word (which is what
// a shared head cost: two thirds of every block's time went into claiming) -- and takes a new one when it has used it. A ready slot waits
// for the holder of its ticket to finish the chunk it is working on; in the meantime the holder works on bounce 0.
struct RpFqClaimState {
    uint32_t ticket;
    bool have_ticket, b0_exhausted;
    uint32_t polls, idle_polls;
};
RP_DEV RpFqWork rp_fq_claim(const RpFrameQueues &fq, RpFqClaimState &cs) {
    RpFqState *const st = fq.st;
    const uint32_t n0 = rp_fq_ld(&st->tail[0]);
    const uint32_t nsl0 = (n0 + RP_CHUNK - 1u) / RP_CHUNK;
    RpFqWork w;
    w.b = ~0u;
    w.slot = w.m = w.n = 0u;
    for (uint32_t idle = 0;;) {
        ++cs.polls;
        // 1. the ready slot of a later bounce that this block's ticket names
        if (fq.n_pub > 1) {
            if (!cs.have_ticket) {
                cs.ticket = rp_fq_add(&st->ring_head, 1u);
                cs.have_ticket = true;
            }
            const unsigned long long e = __hip_atomic_load(fq.ring + cs.ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(e >> 48) == fq.epoch) {
                cs.have_ticket = false;
                w.b = (uint32_t)(e >> 44) & 15u;
                w.slot = (uint32_t)(e >> 12);
                w.m = 1u;
                w.n = (uint32_t)e & 4095u;
                return w;
            }
        }
        // 2. the next slots of bounce 0
        if (!cs.b0_exhausted) {
            const uint32_t s = rp_fq_add(&st->head0, (uint32_t)fq.k0);
            if (s < nsl0) {
                w.b = 0u;
                w.slot = s;
                w.m = min((uint32_t)fq.k0, nsl0 - s);
                w.n = min(w.m * RP_CHUNK, n0 - s * RP_CHUNK);
                return w;
            }
            cs.b0_exhausted = true;
        }
        // 3. nothing to do right now: the ticket's entry is this block's own word to watch
        if ((idle & 3u) == 0u && rp_fq_ld(&st->complete) != 0u) return w;
        ++cs.idle_polls;
        if (++idle > (1u << 21) || ((idle & 255u) == 0u && rp_fq_ld(&st->timeout) != 0u)) { // seconds without work while the frame is not complete: never on a correct run
            rp_fq_add(&st->timeout, 1u);
            return w;
        }
        __builtin_amdgcn_s_sleep(64);
    }
}
// all threads of the block: appends n ids (LDS) to bounce b's queue. The caller's waves have drained their path-state stores and a barrier
// lies behind them.
RP_DEV void rp_fq_publish(const RpFrameQueues &fq, uint32_t b, const uint32_t *ids, uint32_t n, uint32_t *s_base) {
    if (n == 0u) return; // (block-uniform)
    if (threadIdx.x == 0) *s_base = rp_fq_add(&fq.st->tail[b], n);
    __syncthreads();
    const uint32_t base = *s_base;
    uint32_t *const q = rp_fq_ids(fq, b);
    for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) rp_st1<true>(q, base + j, ids[j]);
    rp_drain_stores(); // every writing wave: the ids have left it before the slot says so
    __syncthreads();
    if (threadIdx.x == 0) {
#ifdef RP_FRAME_PROF
        const long long tf0 = wall_clock64();
#endif
        if (!(fq.dbg & 1u)) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // path state + ids of the whole block are in memory ...
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... (the compiler may drop the wait behind buffer_wbl2: restated)
        }
#ifdef RP_FRAME_PROF
        atomicAdd(&fq.st->t_fence, (unsigned long long)(wall_clock64() - tf0));
#endif
        const uint32_t s0 = base / RP_CHUNK, n0 = min(n, (s0 + 1u) * RP_CHUNK - base);
        // the producer that completes a slot hands it on (every other producer's entries were released before its own add)
        if (rp_fq_add(rp_fq_commit(fq, b) + s0, n0) + n0 == RP_CHUNK) rp_fq_push(fq, b, s0, RP_CHUNK);
        if (n > n0 && rp_fq_add(rp_fq_commit(fq, b) + s0 + 1u, n - n0) + (n - n0) == RP_CHUNK) rp_fq_push(fq, b, s0 + 1u, RP_CHUNK);
    }
}
template <int VARIANT, bool LIGHTS, bool TEX, bool ALPHA, bool SINGLE, bool TABLE>
__global__ __launch_bounds__(256, RP_FRAME_WAVES) void rp_k_frame(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpFrameQueues fq, RpCounters *ctr,
                                                                 int *gstack) {
    // one arena for the phases that take turns: the traversal stacks of extend / connect, the shade phase's scratch
    constexpr uint32_t STACK_WORDS = RP_LDS_STACK * RP_TRAVERSE_BLOCK;
    constexpr uint32_t SCRATCH_WORDS = RP_CHUNK + (LIGHTS ? RP_SHADE_RIS_REQ_FLOATS + RP_SHADE_RIS_CONTRIB_FLOATS : 0);
    __shared__ __attribute__((aligned(16))) uint32_t s_arena[STACK_WORDS > SCRATCH_WORDS ? STACK_WORDS : SCRATCH_WORDS];
    __shared__ uint32_t s_ids[RP_CHUNK];    // the list being processed; the shade phase leaves its survivors here (it has read the list by then)
    __shared__ uint32_t s_shadow[RP_CHUNK]; // the shadow rays of the chunk
    __shared__ uint32_t s_cursor[2];
    __shared__ RpFqWork s_work;
    __shared__ uint32_t s_base;
    int *const stack = reinterpret_cast<int *>(s_arena);
    RpShadeLds lds;
    lds.next = s_ids;
    lds.shadow = s_shadow;
    lds.list = s_arena;
    lds.ris_req = reinterpret_cast<float *>(s_arena + RP_CHUNK);
    lds.ris_contrib = reinterpret_cast<float *>(s_arena + RP_CHUNK + RP_SHADE_RIS_REQ_FLOATS);
    RpFqClaimState cs; // (lane 0)
    cs.ticket = cs.polls = cs.idle_polls = 0u;
    cs.have_ticket = cs.b0_exhausted = false;
#ifdef RP_FRAME_PROF
    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const long long pt_begin = wall_clock64();
#define RP_FP_T0 const long long pt0_ = wall_clock64();
#define RP_FP_T1(k) pt[k] += wall_clock64() - pt0_;
#else
#define RP_FP_T0
#define RP_FP_T1(k)
#endif
    for (;;) {
        __syncthreads(); // the previous slot's readers of s_work / s_ids are done
        {
            RP_FP_T0
            if (threadIdx.x == 0) s_work = rp_fq_claim(fq, cs);
            __syncthreads();
            RP_FP_T1(0)
        }
        const RpFqWork w = s_work;
        if (w.b == ~0u) break;
        uint32_t b = w.b, n = w.n;
        const uint32_t base = w.slot * RP_CHUNK; // bounce 0: the path id of the first entry
        if (b > 0u)
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) {
                uint32_t id = rp_ld1<true>(rp_fq_ids(fq, b), base + j);
                if (id >= fq.capacity) { // (a protocol error would otherwise show up as a wild store)
                    rp_fq_add(&fq.st->bad_ids, 1u);
                    id = 0u;
                }
                s_ids[j] = id;
            }
        for (;;) { // the slot's paths, bounce by bounce while they stay with this block
            if (threadIdx.x < 2) s_cursor[threadIdx.x] = 0;
            __syncthreads();
            {
                RP_FP_T0
                if (b == 0u)
                    rp_extend_body<false, true, ALPHA, SINGLE, true, TABLE, 0, true, true>(sc, f, ps, nullptr, n, &s_cursor[0], ctr, gstack, base, stack);
                else
                    rp_extend_body<false, false, ALPHA, SINGLE, true, TABLE, 0, true, true>(sc, f, ps, s_ids, n, &s_cursor[0], ctr, gstack, 0u, stack);
                rp_drain_stores(); // (the alpha test's generator goes to the shade phase through the path state)
                __syncthreads();
                RP_FP_T1(1)
            }
            const bool publish = int(b) + 1 < fq.n_pub;
            uint32_t n_next = 0;
            for (uint32_t sub = 0; sub * RP_CHUNK < n; ++sub) { // (more than one pass only for k0 > 1 slots of bounce 0, which are always published)
                const uint32_t cnt = min((uint32_t)RP_CHUNK, n - sub * RP_CHUNK);
                uint32_t *next = nullptr, *shadow = nullptr;
                uint32_t n_shadow = 0;
                RP_FP_T0
                if (b == 0u)
                    rp_shade_body<VARIANT, true, LIGHTS, TEX, true, TABLE, true, true>(sc, f, ps, sq, nullptr, cnt, nullptr, nullptr, nullptr, ctr, next, n_next, shadow,
                                                                                       n_shadow, base + sub * RP_CHUNK, &lds);
                else
                    rp_shade_body<VARIANT, false, LIGHTS, TEX, true, TABLE, true, true>(sc, f, ps, sq, s_ids, cnt, nullptr, nullptr, nullptr, ctr, next, n_next, shadow,
                                                                                        n_shadow, 0u, &lds);
                rp_drain_stores(); // the shadow rays' contributions are added to the radiance this phase stored
                __syncthreads();
                RP_FP_T1(2)
                {
                    RP_FP_T0
                    if (threadIdx.x == 0) s_cursor[1] = 0;
                    __syncthreads();
                    rp_connect_body<false, ALPHA, SINGLE, true, 0, true, true>(sc, f, ps, sq, s_shadow, n_shadow, &s_cursor[1], ctr, gstack, stack);
                    rp_drain_stores(); // every wave: its path-state stores are in memory before the survivors are handed on
                    __syncthreads();
                    RP_FP_T1(3)
                }
                if (publish) {
                    RP_FP_T0
                    rp_fq_publish(fq, b + 1u, s_ids, n_next, &s_base);
                    __syncthreads();
                    RP_FP_T1(4)
                }
            }
            if (publish || n_next == 0u || int(b) + 1 >= f.rp.max_path_depth) break;
            ++b; // the survivors stay with this block (s_ids holds them)
            n = n_next;
        }
        if (threadIdx.x == 0) {
            rp_drain_stores(); // the commits (and pushes) of this slot's survivors first
            rp_fq_add(&fq.st->done[w.b], w.m);
            rp_drain_stores();
            rp_fq_try_finalize(fq, w.b + 1u);
        }
    }
    if (threadIdx.x == 0) {
        rp_fq_add(&fq.st->polls, cs.polls);
        rp_fq_add(&fq.st->idle_polls, cs.idle_polls);
#ifdef RP_FRAME_PROF
        atomicAdd(&fq.st->t_claim, (unsigned long long)pt[0]);
        atomicAdd(&fq.st->t_extend, (unsigned long long)pt[1]);
        atomicAdd(&fq.st->t_shade, (unsigned long long)pt[2]);
        atomicAdd(&fq.st->t_connect, (unsigned long long)pt[3]);
        atomicAdd(&fq.st->t_publish, (unsigned long long)pt[4]);
        atomicAdd(&fq.st->t_total, (unsigned long long)(wall_clock64() - pt_begin));
#endif
    }
#undef RP_FP_T0
#undef RP_FP_T1
}

// ------------------------------------------------------------------ stream: the frame as TWO co-resident persistent kernels
// rp_k_frame (above) shows what one kernel cannot do: its blocks carry the registers of the shade phase through their traversal phases (4
// waves per SIMD at most) and refill their lanes from block-local pools -- a frame alone takes 2.6 ms where the stage launches take 2.0.
// What frames in flight have -- traversal waves that refill from long queues at five per SIMD, with shade work of other frames in the
// issue slots they leave -- needs kernels with their own register budgets side by side. So: a TRACER kernel (persistent waves, <= 96
// VGPRs, four blocks per CU) and a SHADER kernel (one block per CU) run for the whole frame and hand each other work through memory.
//   items     an item is a path that left a shade (or a camera path): an optional shadow ray, then an optional continuation ray, traced one
//             after the other by ONE lane (dstream.h), so that the shadow ray's contribution is added to the path's radiance before the
//             next shade adds anything -- the megakernel's order of additions with no dependency between lanes.
//   S0        the camera paths: the identity over the path ids, dealt to tracer waves in pools of RP_ST_POOL entries (head: s0_head)
//   R         the items the shader kernel emits (u32: path id | flags), appended with one atomicAdd per chunk of survivors (r_tail)
//   chunks    RP_CHUNK consecutive entries of S0 or R are the unit of shading: traced[chunk] counts the entries whose rays are done (the
//             tracer wave whose add completes a chunk hands it to the shader ring), commit[chunk] the entries of an R chunk that have
//             been written (the shader block whose add completes it hands its four pools to the tracer ring)
//   rings     64-bit entries tagged with the frame's epoch; a consumer (tracer wave / shader block) holds a TICKET -- position in the ring
//             that is its own to watch -- so nobody races for a head word (as in rp_k_frame)
//   the end   only shader blocks append. When a shader block finishes a chunk and finds every chunk that was ever handed out shaded
//             (shaded == S0 chunks + R chunks handed to the tracers), nothing is in flight and R's tail stands still: it SEALS the last,
//             partly filled chunk (pads it with RP_ITEM_NONE, which completes it) -- or, if there is none, sets `complete`.
// Visibility: as rp_k_frame -- consumers read around the L1 (sc1), a producer drains its stores and issues ONE agent-scope release before
// the atomic that publishes (a shader block per chunk; a tracer wave per report of finished items).
#define RP_ST_POOL 256u // entries a tracer wave takes at a time
#ifndef RP_ST_REPORT
#define RP_ST_REPORT 192u     // ended items of one chunk a tracer wave collects before it reports them
#endif
#ifndef RP_ST_REPORT_AGE
#define RP_ST_REPORT_AGE 24u  // ... or this many refill rounds, whichever comes first
#endif
struct RpStState {
    uint32_t r_tail, s0_head, tr_tail, tr_head, sr_tail, sr_head;
    uint32_t tr_chunks; // R chunks handed to the tracer ring so far
    uint32_t shaded;    // chunks shaded
    uint32_t complete;
    uint32_t timeout, overflow, seals;
    uint32_t n0;        // camera paths (preset by the host)
    uint32_t trace_polls, shade_polls, _pad;
    unsigned long long t_wait, t_load, t_shade, t_append, t_fence, t_total, t_trace_idle, t_trace_total, t_trace_report; // -DRP_FRAME_PROF: 100 MHz ticks (lane 0 of a block / wave)
};
struct RpStream {
    RpStState *st;
    uint32_t *r;                 // the items of later bounces
    uint32_t *commit;            // per R chunk
    uint32_t *traced;            // per chunk: S0 chunks first, R chunks behind them
    unsigned long long *tr_ring; // pools of R for tracer waves: epoch << 48 | first entry
    unsigned long long *sr_ring; // chunks for shader blocks: epoch << 48 | chunk id (S0 chunks first)
    uint32_t r_capacity;         // entries R holds (a multiple of RP_CHUNK)
    uint32_t n_s0_chunks;
    uint32_t epoch;
    uint32_t capacity;           // path ids are below this
};
RP_DEV void rp_st_push(unsigned long long *ring, uint32_t *tail, uint32_t epoch, uint32_t value, uint32_t n = 1u, uint32_t step = 0u) {
    const uint32_t i = rp_fq_add(tail, n);
    for (uint32_t k = 0; k < n; ++k)
        __hip_atomic_store(ring + i + k, ((unsigned long long)epoch << 48) | (unsigned long long)(value + k * step), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
RP_DEV uint32_t rp_st_chunk_size(const RpStream &sx, uint32_t chunk, uint32_t n0) {
    return chunk < sx.n_s0_chunks ? min((uint32_t)RP_CHUNK, n0 - chunk * RP_CHUNK) : (uint32_t)RP_CHUNK;
}

// ---- the tracer: persistent waves over items
template <bool SINGLE, bool TABLE>
__global__ RP_TRAVERSE_BOUNDS void rp_k_stream_trace(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpStream sx, int *gstack) {
    RpStState *const st = sx.st;
    const uint32_t lane = rp_lane_id();
    const uint32_t n0 = rp_fq_ld(&st->n0);
    // wave-uniform consumer state
    uint32_t ticket = 0;
    bool have_ticket = false, s0_done = false;
    uint32_t idle_polls = 0;
    // the lane's item
    uint32_t my_p = 0, my_flags = 0, my_chunk = 0, fin_chunk = RP_ITEM_NONE;
    auto pool = [&](uint32_t &first, uint32_t &end) -> int {
        int r = 0;
        uint32_t a = 0, b = 0;
        if (lane == 0) {
            if (!have_ticket) {
                ticket = rp_fq_add(&st->tr_head, 1u);
                have_ticket = true;
            }
            const unsigned long long e = __hip_atomic_load(sx.tr_ring + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(e >> 48) == sx.epoch) {
                have_ticket = false;
                a = 0x80000000u | (uint32_t)e;
                b = a + RP_ST_POOL;
                r = 1;
            } else if (!s0_done) {
                const uint32_t s = rp_fq_add(&st->s0_head, RP_ST_POOL);
                if (s < n0) {
                    a = s;
                    b = min(n0, s + RP_ST_POOL);
                    r = 1;
                } else
                    s0_done = true;
            }
            if (r == 0) {
                if (rp_fq_ld(&st->complete) != 0u) r = -1;
                else if (++idle_polls > (1u << 22) || ((idle_polls & 1023u) == 0u && rp_fq_ld(&st->timeout) != 0u)) {
                    rp_fq_add(&st->timeout, 1u);
                    r = -1;
                }
            } else
                idle_polls = 0;
        }
        r = __builtin_amdgcn_readfirstlane(r);
        first = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
        end = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
        return r;
    };
    auto cont_ray = [&](bool first, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool { // the closest-hit ray of the lane's item
        if (first) {
            RpRng rng;
            rd = v3(0.0f, 0.0f, 1.0f);
            if (!rp_primary_ray<TABLE>(f, my_p, rng, rd, ro)) return false; // tile padding: no such pixel sample
            tmin = 0.0f;
            tmax = 2.e32f;
        } else {
            const float4 o4 = rp_ld4<true>(ps.ray_o, my_p), d4 = rp_ld4<true>(ps.ray_d, my_p);
            ro = xyz(o4);
            rd = xyz(d4);
            tmin = rp_geometry_scale_to_tmin(ro, o4.w);
            tmax = 1e20f;
        }
        return true;
    };
    auto begin = [&](uint32_t idx, V3 &ro, V3 &rd, float &tmin, float &tmax, bool &anyq) -> bool {
        if (idx & 0x80000000u) {
            const uint32_t ri = idx & 0x7FFFFFFFu;
            const uint32_t e = rp_ld1<true>(sx.r, ri);
            my_chunk = sx.n_s0_chunks + ri / RP_CHUNK;
            if (e == RP_ITEM_NONE || (e & RP_ITEM_PATH) >= sx.capacity) { // padding of a sealed chunk (or, never on a correct run, no path id)
                if (e != RP_ITEM_NONE) rp_fq_add(&st->overflow, 1u);
                fin_chunk = my_chunk;
                return false;
            }
            my_p = e & RP_ITEM_PATH;
            my_flags = e & (RP_ITEM_CONT | RP_ITEM_SHADOW);
            if (my_flags & RP_ITEM_SHADOW) {
                const float4 o4 = rp_ld4<true>(ps.ray_o, my_p), d4 = rp_ld4<true>(sq.d, my_p);
                ro = xyz(o4);
                rd = xyz(d4);
                tmin = rp_geometry_scale_to_tmin(ro, o4.w);
                tmax = d4.w;
                anyq = true;
                return true;
            }
            anyq = false;
            return cont_ray(false, ro, rd, tmin, tmax); // (always true)
        }
        my_p = idx;
        my_chunk = idx / RP_CHUNK;
        my_flags = RP_ITEM_CONT;
        anyq = false;
        if (!cont_ray(true, ro, rd, tmin, tmax)) {
            rp_st2<true>(reinterpret_cast<float2 *>(ps.hit_ids), my_p, make_float2(__int_as_float(-1), __int_as_float(-1))); // a miss is recorded
            fin_chunk = my_chunk;
            return false;
        }
        return true;
    };
    auto next = [&](const RpHitRec &h, V3 &ro, V3 &rd, float &tmin, float &tmax, bool &anyq) -> bool {
        if (anyq) { // the shadow ray: its contribution arrives when nothing is hit (nee.glsl:76-84)
            if (h.inst_idx < 0) {
                const float4 c = rp_ld4<true>(sq.contrib, my_p);
                float4 il = rp_ld4<true>(ps.illum, my_p);
                il.x += c.x;
                il.y += c.y;
                il.z += c.z;
                rp_st4<true>(ps.illum, my_p, il);
            }
            if (my_flags & RP_ITEM_CONT) {
                anyq = false;
                return cont_ray(false, ro, rd, tmin, tmax);
            }
            fin_chunk = my_chunk;
            return false;
        }
        rp_st4<true>(ps.hit_tuv, my_p, make_float4(h.t, h.u, h.v, __int_as_float(h.prim)));
        rp_st2<true>(reinterpret_cast<float2 *>(ps.hit_ids), my_p, make_float2(__int_as_float(h.inst_idx), __int_as_float(h.geom)));
        fin_chunk = my_chunk;
        return false;
    };
    // Ended items are counted per wave in a few (chunk, count) slots and handed over -- a release fence + one atomicAdd per slot -- when a slot
    // holds RP_ST_REPORT items, when the wave runs out of slots, when it has nothing in flight, or after RP_ST_REPORT_AGE steps: a report per
    // refill (every ~48 items) was half a million L2 write-backs per frame and made the frame six times slower
    uint32_t slot_chunk[4] = {RP_ITEM_NONE, RP_ITEM_NONE, RP_ITEM_NONE, RP_ITEM_NONE}, slot_count[4] = {0u, 0u, 0u, 0u}, slot_age = 0u; // wave-uniform
    auto flush = [&](uint32_t which_mask) { // wave-uniform
        rp_drain_stores(); // this wave's hit records and radiance updates have left it ...
        if (lane == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // ... and are in memory before a chunk is called done
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if ((which_mask >> k) & 1u) {
                if (lane == 0 && slot_count[k] > 0u &&
                    rp_fq_add(sx.traced + slot_chunk[k], slot_count[k]) + slot_count[k] == rp_st_chunk_size(sx, slot_chunk[k], n0))
                    rp_st_push(sx.sr_ring, &st->sr_tail, sx.epoch, slot_chunk[k]);
                slot_chunk[k] = RP_ITEM_NONE;
                slot_count[k] = 0u;
            }
    };
    auto report = [&](bool now, bool idle_wave) {
        unsigned long long mask = __ballot(fin_chunk != RP_ITEM_NONE);
        while (mask != 0ull) { // (one or two distinct chunks)
            const int leader = __ffsll((long long)mask) - 1;
            const uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)fin_chunk, leader);
            const unsigned long long same = __ballot(fin_chunk == c);
            const uint32_t cnt = (uint32_t)__popcll(same);
            int k = slot_chunk[0] == c ? 0 : slot_chunk[1] == c ? 1 : slot_chunk[2] == c ? 2 : slot_chunk[3] == c ? 3 : -1;
            if (k < 0) {
                k = slot_chunk[0] == RP_ITEM_NONE ? 0 : slot_chunk[1] == RP_ITEM_NONE ? 1 : slot_chunk[2] == RP_ITEM_NONE ? 2 : slot_chunk[3] == RP_ITEM_NONE ? 3 : -1;
                if (k < 0) { // no slot left: hand everything over
                    flush(15u);
                    k = 0;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j == k) {
                    slot_chunk[j] = c;
                    slot_count[j] += cnt;
                }
            if (fin_chunk == c) fin_chunk = RP_ITEM_NONE;
            mask &= ~same;
        }
        uint32_t due = 0u;
        bool pending = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (slot_count[j] >= RP_ST_REPORT) due |= 1u << j;
            pending = pending || slot_count[j] > 0u;
        }
        if (!pending) {
            slot_age = 0u;
            return;
        }
        if (now && ++slot_age >= RP_ST_REPORT_AGE) due = 15u;
        if (idle_wave) due = 15u; // nothing in flight: whoever waits for these chunks should not wait for this wave's next refill
        if (due != 0u) {
            flush(due);
            slot_age = 0u;
        }
    };
    rp_wave_trace_items<false, SINGLE>(sc, gstack, pool, begin, next, report, RpNoAlpha());
}

// ---- the shader: persistent blocks over chunks
template <int VARIANT, bool LIGHTS, bool TEX, bool TABLE>
__global__ __launch_bounds__(256, RP_SHADE_WAVES) void rp_k_stream_shade(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpStream sx, RpCounters *ctr) {
    __shared__ uint32_t s_ids[RP_CHUNK], s_shadow_unused[1];
    __shared__ uint32_t s_list[RP_CHUNK];
    __shared__ float s_ris_req[LIGHTS ? RP_SHADE_RIS_REQ_FLOATS : 1], s_ris_contrib[LIGHTS ? RP_SHADE_RIS_CONTRIB_FLOATS : 1];
    __shared__ uint32_t s_chunk, s_n, s_base, s_seal;
    RpStState *const st = sx.st;
    RpShadeLds lds;
    lds.next = s_ids;
    lds.shadow = s_shadow_unused;
    lds.list = s_list;
    lds.ris_req = s_ris_req;
    lds.ris_contrib = s_ris_contrib;
#ifdef RP_STREAM_DEBUG
    if (blockIdx.x == 0 && threadIdx.x == 0)
        printf("[dev shade] st %p r %p commit %p traced %p tr %p sr %p cap %u chunks0 %u epoch %u pathcap %u\n", (void *)sx.st, (void *)sx.r, (void *)sx.commit, (void *)sx.traced,
               (void *)sx.tr_ring, (void *)sx.sr_ring, sx.r_capacity, sx.n_s0_chunks, sx.epoch, sx.capacity);
#endif
    const uint32_t n0 = rp_fq_ld(&st->n0);
    uint32_t ticket = 0, idle = 0; // (thread 0)
    bool have_ticket = false;
    // all threads: appends n entries (LDS) to R; the chunks it completes go to the tracers
    auto append = [&](const uint32_t *entries, uint32_t n, bool none) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // the path state and shadow rays of this chunk are in memory before its items are
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            s_base = rp_fq_add(&st->r_tail, n);
        }
        __syncthreads();
        const uint32_t base = s_base;
        if (base + n > sx.r_capacity) { // (never with the capacity the host allocates: total items of a frame; reported, the frame is cut short)
            if (threadIdx.x == 0) {
                rp_fq_add(&st->overflow, 1u);
                __hip_atomic_store(&st->complete, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            return;
        }
        for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) rp_st1<true>(sx.r, base + j, none ? RP_ITEM_NONE : entries[j]);
        rp_drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (uint32_t c = base / RP_CHUNK; c * RP_CHUNK < base + n; ++c) {
                const uint32_t lo = max(base, c * RP_CHUNK), hi = min(base + n, (c + 1u) * RP_CHUNK);
                if (rp_fq_add(sx.commit + c, hi - lo) + (hi - lo) == RP_CHUNK) {
                    rp_fq_add(&st->tr_chunks, 1u);
                    rp_drain_stores();
                    rp_st_push(sx.tr_ring, &st->tr_tail, sx.epoch, c * RP_CHUNK, RP_CHUNK / RP_ST_POOL, RP_ST_POOL);
                }
            }
        }
    };
#ifdef RP_FRAME_PROF
    long long pt[5] = {0, 0, 0, 0, 0};
    const long long pt_begin = wall_clock64();
    long long pt0 = pt_begin;
#define RP_SP(k) { const long long t_ = wall_clock64(); pt[k] += t_ - pt0; pt0 = t_; }
#else
#define RP_SP(k)
#endif
    for (;;) {
        __syncthreads();
        RP_SP(3)
        if (threadIdx.x == 0) { // the next chunk: this block's ticket of the shader ring
            uint32_t chunk = RP_ITEM_NONE;
            for (;;) {
                if (!have_ticket) {
                    ticket = rp_fq_add(&st->sr_head, 1u);
                    have_ticket = true;
                }
                const unsigned long long e = __hip_atomic_load(sx.sr_ring + ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((uint32_t)(e >> 48) == sx.epoch) {
                    have_ticket = false;
                    chunk = (uint32_t)e;
                    idle = 0;
                    break;
                }
                if (rp_fq_ld(&st->complete) != 0u) break;
                if (++idle > (1u << 21) || ((idle & 255u) == 0u && rp_fq_ld(&st->timeout) != 0u)) {
                    rp_fq_add(&st->timeout, 1u);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
            s_chunk = chunk;
            s_n = 0;
        }
        __syncthreads();
        const uint32_t chunk = s_chunk;
        RP_SP(0)
        if (chunk == RP_ITEM_NONE) break;
        const bool first = chunk < sx.n_s0_chunks;
        uint32_t n = 0, base = 0;
        if (first) {
            base = chunk * RP_CHUNK;
            n = min((uint32_t)RP_CHUNK, n0 - base);
        } else { // the paths of the chunk that go on: their continuation rays have been traced
            const uint32_t r0 = (chunk - sx.n_s0_chunks) * RP_CHUNK;
            for (uint32_t j = threadIdx.x; j < RP_CHUNK; j += blockDim.x) {
                const uint32_t e = rp_ld1<true>(sx.r, r0 + j);
                const bool go = e != RP_ITEM_NONE && (e & RP_ITEM_CONT) != 0u;
                const uint32_t at = rp_wave_append(&s_n, go);
                if (go) s_ids[at] = e & RP_ITEM_PATH;
            }
            __syncthreads();
            n = s_n;
        }
        uint32_t *next = nullptr, *shadow = nullptr;
        uint32_t n_next = 0, n_shadow = 0;
        RP_SP(1)
        if (n > 0u) {
            if (first)
                rp_shade_body<VARIANT, true, LIGHTS, TEX, true, TABLE, true, true, true>(sc, f, ps, sq, nullptr, n, nullptr, nullptr, nullptr, ctr, next, n_next, shadow, n_shadow,
                                                                                         base, &lds);
            else
                rp_shade_body<VARIANT, false, LIGHTS, TEX, true, TABLE, true, true, true>(sc, f, ps, sq, s_ids, n, nullptr, nullptr, nullptr, ctr, next, n_next, shadow,
                                                                                          n_shadow, 0u, &lds);
        }
        rp_drain_stores();
        __syncthreads();
        RP_SP(2)
        if (n_next > 0u) append(s_ids, n_next, false);
        __syncthreads();
        // this chunk is shaded. Is it the last one that was in flight?
        if (threadIdx.x == 0) {
            rp_drain_stores();
            const uint32_t sh = rp_fq_add(&st->shaded, 1u) + 1u;
            rp_drain_stores();
            uint32_t seal = 0u;
            if (sh == sx.n_s0_chunks + rp_fq_ld(&st->tr_chunks)) { // every chunk ever handed out is shaded: nobody appends any more
                const uint32_t t = rp_fq_ld(&st->r_tail);
                if (t % RP_CHUNK != 0u) {
                    seal = RP_CHUNK - t % RP_CHUNK;
                    rp_fq_add(&st->seals, 1u);
                } else
                    __hip_atomic_store(&st->complete, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            s_seal = seal;
        }
        __syncthreads();
        if (s_seal > 0u) append(nullptr, s_seal, true); // pads the last chunk: that completes it and sends it to the tracers
    }
#ifdef RP_FRAME_PROF
    if (threadIdx.x == 0) {
        atomicAdd(&st->t_wait, (unsigned long long)pt[0]);
        atomicAdd(&st->t_load, (unsigned long long)pt[1]);
        atomicAdd(&st->t_shade, (unsigned long long)pt[2]);
        atomicAdd(&st->t_append, (unsigned long long)pt[3]);
        atomicAdd(&st->t_total, (unsigned long long)(wall_clock64() - pt_begin));
    }
#endif
#undef RP_SP
}

// rp_kernarg (above) reads RpScene / RpFrame at the offsets they have as the FIRST TWO by-value arguments of a kernel: both kernels that run
// rp_shade_body must start their argument lists that way.
template <class F>
struct rp_args_start_with_scene_and_frame : std::false_type {};
template <class... Rest>
struct rp_args_start_with_scene_and_frame<void (*)(RpScene, RpFrame, Rest...)> : std::true_type {};
static_assert(rp_args_start_with_scene_and_frame<decltype(&rp_k_shade<RPTR_VARIANT_SIMPLE, true, false, false, false>)>::value &&
                  rp_args_start_with_scene_and_frame<decltype(&rp_k_tail<RPTR_VARIANT_SIMPLE, false, false, false, true, false>)>::value &&
                  rp_args_start_with_scene_and_frame<decltype(&rp_k_frame<RPTR_VARIANT_SIMPLE, false, false, false, true, false>)>::value &&
                  rp_args_start_with_scene_and_frame<decltype(&rp_k_stream_shade<RPTR_VARIANT_SIMPLE, false, false, false>)>::value,
              "rp_k_shade / rp_k_tail / rp_k_frame / rp_k_stream_shade: (RpScene, RpFrame, ...) must come first (kernels.h rp_kernarg)");
