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
    float4 *hit_tuv; // t, u, v of the closest hit (.w unused)
    int2 *hit_ids;   // index of the hit instance record (-1 = miss), index of the hit triangle in RpScene::tris (= of its shading record, dshade.h RpShadeTri)
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
    dir = norm3_ieee(point.x * du + point.y * dv + tl); // (IEEE in both builds of the shading stages: the first extend and a first shade that makes the ray again agree)
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

// Round 5: the first extend STORES the camera ray's direction and generator state (ray_d[p], 16 bytes) and the first shade loads them instead
// of making the camera ray a second time (generator seeding, the pixel jitter, an IEEE normalisation, the tile -> pixel arithmetic: ~150 of the
// 910 instructions a first-bounce shade wave executes; the pipeline is bound by instruction issue, HBM carries 0.17 of its peak) -- the same
// float bits either way. Not for table point sets (TABLE: the generator's other fields would have to travel too). -DRP_FIRST_RAY_STORED=0: as before.
#ifndef RP_FIRST_RAY_STORED
#define RP_FIRST_RAY_STORED 1
#endif
// hit_ids.x of a slot of the first queue that names no pixel sample (tile padding): neither a hit nor a miss
#define RP_HIT_PADDING (-2)

// ------------------------------------------------------------------ extend (closest hit), persistent waves
// FIRST: bounce 0, the rays are the camera rays (computed, not loaded)
// ALPHA: the scene has alpha-tested materials. The test of a candidate may draw from the path's generator
// (pt_megakernel.glsl:354-358), so the lane carries it through the traversal and hands it back in the path state.
// SINGLE: the scene has one instance record; queries start inside it (dtraverse.h).
// LOCAL: `queue` / `cursor` are a block-local list and its cursor in LDS (rp_k_tail).
// EXTLDS (rp_k_tail): the stacks' LDS belongs to the caller.
template <bool COUNT, bool FIRST, bool ALPHA, bool SINGLE, bool LOCAL, bool TABLE, int LDSTOP = 0, bool EXTLDS = false>
RP_DEV void rp_extend_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const uint32_t *queue, uint32_t n, uint32_t *cursor, RpCounters *ctr,
                           int *gstack, int *ext_stack = nullptr) {
    uint32_t n_nodes = 0, n_tris = 0;
    uint32_t lane_rng = 0, lane_rng_in = 0; // ALPHA only
    uint32_t lane_p = 0; // the path whose ray this lane traces
    auto load = [&](uint32_t i, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool {
        const uint32_t p = (FIRST && !queue) ? i : queue[i]; // FIRST: the first queue is the identity (NULL) unless the caller stored it
        lane_p = p;
        if (FIRST) {
            RpRng rng;
            rd = v3(0.0f, 0.0f, 1.0f);
            if (!rp_primary_ray<TABLE>(f, p, rng, rd, ro)) { // tile padding: no such pixel sample (the first shade's regrouping pass skips it)
                ps.hit_ids[p] = make_int2(RP_HIT_PADDING, -1);
                return false;
            }
            if (RP_FIRST_RAY_STORED && !TABLE) ps.ray_d[p] = f4(rd, __uint_as_float(rng.s)); // (an alpha test that draws from the generator stores its new state in done())
            tmin = 0.0f;
            tmax = 2.e32f;
            if (ALPHA) lane_rng = (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) ? rng.s : rp_alpha_seed(f, p);
        } else {
            const float4 o = ps.ray_o[p], d = ps.ray_d[p];
            ro = xyz(o);
            rd = xyz(d);
            tmin = rp_geometry_scale_to_tmin(ro, o.w); // (what the shade computed for its shadow ray: same operands, same bits)
            tmax = 1e20f;
            if (ALPHA)
                lane_rng = lane_rng_in = (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) ? __float_as_uint(d.w) : ps.alpha_rng[p];
        }
        return true;
    };
    auto done = [&](uint32_t, const RpHitRec &h) {
        const uint32_t p = lane_p;
        if (h.inst_idx >= 0) ps.hit_tuv[p] = make_float4(h.t, h.u, h.v, 0.0f); // (nobody reads t / u / v of a miss: 16 bytes less per ray that leaves the scene)
        ps.hit_ids[p] = make_int2(h.inst_idx, h.tri);
        if (ALPHA) {
            if (TABLE && f.rng_variant != RPTR_RNG_VARIANT_UNIFORM) {
                if (FIRST || lane_rng != lane_rng_in) ps.alpha_rng[p] = lane_rng;
            } else if (FIRST)
                (reinterpret_cast<uint32_t *>(ps.ray_d))[4u * p + 3u] = lane_rng; // the first shade takes it from here (f.alpha_test)
            else if (lane_rng != lane_rng_in)
                (reinterpret_cast<uint32_t *>(ps.ray_d))[4u * p + 3u] = lane_rng;
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
template <bool COUNT, bool ALPHA, bool SINGLE, bool LOCAL, int LDSTOP = 0, bool EXTLDS = false>
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
        const float4 o = ps.ray_o[p], d = sq.d[p]; // the vertex: origin of the shadow ray and of the continuation ray
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
            float4 il = ps.illum[p];
            il.x += c.x;
            il.y += c.y;
            il.z += c.z;
            ps.illum[p] = il;
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
// the LDS buffers of rp_shade_body when the caller owns them (EXTLDS; rp_k_tail): `next` and `shadow` (RP_CHUNK words each) outlive the call
// (the survivors and the shadow rays of the chunk), `list` (RP_CHUNK words), `ris_req` (2048 floats) and `ris_contrib` (4096 floats; LIGHTS
// only) are scratch the caller may reuse between calls
struct RpShadeLds {
    uint32_t *next, *shadow, *list;
    float *ris_req, *ris_contrib;
};
#define RP_SHADE_RIS_REQ_FLOATS ((256 / 64) * 64 * 8)
#define RP_SHADE_RIS_CONTRIB_FLOATS ((256 / 64) * 64 * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE)
// EXTLDS: as for rp_extend_body
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX, bool LOCAL, bool TABLE, bool EXTLDS = false>
RP_DEV void rp_shade_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq, const uint32_t *order, const uint32_t n,
                          uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count, RpCounters *ctr, uint32_t *&local_next, uint32_t &n_next,
                          uint32_t *&local_shadow, uint32_t &n_shadow, const RpShadeLds *ext = nullptr) {
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
            bool valid = i < n;
            uint32_t pp = 0;
            bool is_hit = false;
            if (valid) {
                pp = (FIRST && !order) ? i : order[i];
                const int hx = ps.hit_ids[pp].x;
                is_hit = hx >= 0;
                if (FIRST && hx == RP_HIT_PADDING) valid = false; // tile padding of the first queue: no pixel sample
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
                const uint32_t key = sc.shade[ps.hit_ids[pp].y].material & 63u;
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
                // bounce 0: the camera ray -- as the first extend stored it (direction, generator state; the origin is the frame's camera), or made
                // again (table point sets). Ids of tile padding were dropped by the regrouping pass.
                if (FIRST && RP_FIRST_RAY_STORED && !TABLE) {
                    const float4 rd4 = ps.ray_d[p];
                    ray_dir = xyz(rd4);
                    rng.s = __float_as_uint(rd4.w);
                    first_sslot = rp_div(p, f.div_npix_padded);
                    ray_origin = ld3(f.cam_pos);
                    const bool need_pixel = f.aov_albedo_roughness != nullptr; // (the AOV images are written by the first sample of the frame)
                    if (f.per_frame_cams != 0 || TEX) {
                        const RpSlotFrame sf0 = rp_slot_frame(f, first_sslot);
                        V3 du = ld3(f.cam_du), dv = ld3(f.cam_dv);
                        if (f.per_frame_cams != 0) {
                            const RpCam &cm = f.cams[min(sf0.frame, (uint32_t)(RP_BATCH_CAMS - 1))];
                            ray_origin = ld3(cm.pos);
                            du = ld3(cm.du);
                            dv = ld3(cm.dv);
                        }
                        first_du_dv[0] = du;
                        first_du_dv[1] = dv;
                    }
                    if (need_pixel) (void)rp_slot_to_local(f, p - first_sslot * uint32_t(f.npix_padded), first_lx, first_ly);
                } else if (FIRST)
                    present = rp_primary_ray_ex<TABLE>(f, p, rng, ray_dir, first_lx, first_ly, first_sslot, ray_origin, TEX ? first_du_dv : nullptr);
            }
            if (present) {
                my_closest++;
                if (FIRST) { // init_shading_sample_state (shading_interface.glsl:20-22)
                    if (f.alpha_test && (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) && !(RP_FIRST_RAY_STORED && !TABLE))
                        rng.s = (reinterpret_cast<const uint32_t *>(ps.ray_d))[4u * p + 3u]; // alpha tests of the first extend may have drawn from it
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
                    const float4 ro4 = ps.ray_o[p], rd4 = ps.ray_d[p];
                    const float4 thr4 = ps.thr[p];
                    const float4 il4 = ps.illum[p];
                    rng = rp_rng_resume<TABLE>(f, p, __float_as_uint(rd4.w));
                    total_t = ro4.w;
                    ray_origin = xyz(ro4);
                    ray_dir = xyz(rd4);
                    throughput = xyz(thr4);
                    illum = xyz(il4);
                    prev_bounce_pdf = thr4.w;
                    bounce = __float_as_int(il4.w);
                    if (TEX) {
                        const float4 fp = ps.footprint[p];
                        tex_fp = M2{v2(fp.x, fp.y), v2(fp.z, fp.w)};
                    }
                }
                // (the regrouping put the chunk's hits first: a miss lane needs no hit record)
                const int2 hid = il >= chunk_hits ? make_int2(-1, -1) : ps.hit_ids[p];
                const int hit_inst = hid.x;
                if (hit_inst < 0) {
                    // miss: pt_megakernel.glsl:480-489
                    illum = illum + throughput * rp_compute_sky_illum(f, ray_dir, prev_bounce_pdf);
                    ps.illum[p] = f4(illum, __int_as_float(bounce));
                    if (FIRST && aov_px >= 0) { // pt_megakernel.glsl:482-487
                        rp_store_geometry_aovs(f, aov_px, v3s(0.0f), v3s(2.e32f), aov_jitter);
                        rp_store_material_aovs(f, aov_px, v3s(0.0f), 1.0f, 1.0f);
                    }
                } else {
                    hit_lane = true;
                    my_hits++;
                    // ---- hit attributes, pt_megakernel.glsl:495-572. Everything the hit needs of its triangle is ONE 64-byte record named by
                    // the hit record itself (dshade.h RpShadeTri): its four loads, the instance's rows and -- one step behind -- the material
                    // are in flight together; nothing waits for a geometry record any more.
                    const float4 hit4 = ps.hit_tuv[p];
                    const uint32_t tri_index = uint32_t(hid.y);
                    const float4 *sp = reinterpret_cast<const float4 *>(sc.shade + tri_index);
                    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3];
                    const float4 *ip = reinterpret_cast<const float4 *>(sc.insts + hit_inst);
                    const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
                    const int4 meta = *reinterpret_cast<const int4 *>(ip + 3);
                    const uint32_t mword = __float_as_uint(s3.w);
                    int material_id = int(mword & RP_SHADE_MATERIAL_MASK);
                    if (meta.w & RP_INST_OWN_MATERIALS) { // (an instance of another parameterized mesh of the same mesh: its own material table)
                        const int *tr = reinterpret_cast<const int *>(sc.tris + tri_index); // prim, geom: words 9 and 10
                        material_id = rp_hit_material_id(sc.geoms[meta.y + tr[10]], uint32_t(tr[9]));
                    }
                    const RptrBaseMaterial mp = sc.materials[material_id];
                    // transpose(mat3(world_to_object)): its columns are the rows of world_to_object
                    const M3 n2w{v3(r0.x, r0.y, r0.z), v3(r1.x, r1.y, r1.z), v3(r2.x, r2.y, r2.z)};
                    const uint64_t qa = uint64_t(__float_as_uint(s2.y)) | (uint64_t(__float_as_uint(s2.z)) << 32), qb = uint64_t(__float_as_uint(s2.w)) | (uint64_t(__float_as_uint(s3.x)) << 32),
                                   qc = uint64_t(__float_as_uint(s3.y)) | (uint64_t(__float_as_uint(s3.z)) << 32);
                    RpHit hit = rp_calc_hit_attributes(v3(s0.x, s0.y, s0.z), v3(s0.w, s1.x, s1.y), v3(s1.z, s1.w, s2.x), qa, qb, qc, (mword & RP_SHADE_HAS_NORMALS) != 0u,
                                                       (mword & RP_SHADE_HAS_UVS) != 0u, material_id, hit4.x, hit4.y, hit4.z, n2w);
                    // :578-580
                    float approx_tri_solid_angle = len3(hit.geo_normal);
                    hit.geo_normal = hit.geo_normal / approx_tri_solid_angle;
                    approx_tri_solid_angle *= rp_fdiv(fabsf(dot3(hit.geo_normal, ray_dir)), hit.dist * hit.dist);
                    // :582-606
                    total_t += hit.dist;
                    const RpTexCoord tc = TEX ? rp_hit_texcoord(hit.uv, tex_fp, ray_dir, hit.geo_normal, hit.tangent, hit.bitangent_l, total_t) : rp_texcoord(hit.uv);
                    geometry_scale = total_t;
                    w_o = -ray_dir;
                    ip_p = ray_origin + hit.dist * ray_dir;
                    gn = hit.geo_normal;
                    nn = hit.normal;
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
                        map_nrm.z = rp_fsqrt(fmaxf(1.0f - map_nrm.x * map_nrm.x - map_nrm.y * map_nrm.y, 0.0f));
                        const V3 t_z = f.sp.normal_z_scale * nn;
                        nn = norm3((t_x * map_nrm.x + t_y * map_nrm.y) + t_z * map_nrm.z);
                    }
                    // :656-668
                    {
                        const float nw = dot3(w_o, nn);
                        const float gnw = dot3(w_o, gn);
                        if (nw * gnw <= 0.0f) {
                            const float blend = rp_fdiv(gnw, gnw - nw);
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
                        const float light_pdf = (1.0f - f.sp.sun_radiance[3]) * rp_frcp(float(f.num_bins) * approx_tri_solid_angle);
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
                            sel_sample.x = rp_fdiv(sel_sample.x - sun_w, 1.0f - sun_w);
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
                    sel_sample.x = rp_fdiv(sel_sample.x, sun_w);
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
                    // rays_shadow counts the reference's visibility QUERY (nee.glsl:72-75 traces before it looks at the BSDF's pdf). The device only
                    // queues a shadow ray when bsdf_pdf >= 0 -- a negative pdf contributes nothing either way -- so the counter would exceed the
                    // traced rays by the negative-pdf samples; neither shipped BSDF returns one (rp_simple_pdf, rp_gltf_wpdf >= 0), so today the
                    // two are equal and RptrStats.rays_shadow is exact.
                    if (needs_ray) my_shadow++;
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
                            ps.ray_d[p] = f4(ray_dir, __uint_as_float(rng.s));
                            ps.thr[p] = f4(throughput, prev_bounce_pdf);
                            if (TEX) ps.footprint[p] = make_float4(tex_fp.c0.x, tex_fp.c0.y, tex_fp.c1.x, tex_fp.c1.y);
                        }
                    }
                }
                // the vertex (ip_p: a surviving path's ray_origin is ip_p) and the path length up to it: what the shadow ray and the continuation
                // ray start from (their offset = rp_geometry_scale_to_tmin of the two, recomputed by connect / extend)
                if (alive || has_shadow) ps.ray_o[p] = f4(ip_p, total_t);
                ps.illum[p] = f4(illum, __int_as_float(bounce));
            }
            {
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
// MATH: the build of the shading arithmetic (dmath.h RP_FAST_MATH) -- part of the kernel's NAME only, so that the IEEE and the fast build of
// one instantiation (k_shade.hip / k_tail.hip compiled twice) are two symbols of the library
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX, bool TABLE, int MATH = RP_FAST_MATH>
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
template <int VARIANT, bool LIGHTS, bool TEX, bool ALPHA, bool SINGLE, bool TABLE, int MATH = RP_FAST_MATH>
__global__ __launch_bounds__(256, 1) void rp_k_tail(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, const uint32_t *queue, RpCounters *ctr,
                                                    int first_bounce, int *gstack) {
    // One arena for the phases that take turns (round 4): the LDS stacks of the closest-hit traversal, the shade phase's
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
            rp_extend_body<false, false, ALPHA, SINGLE, true, TABLE, 0, true>(sc, f, ps, s_cur, n, &s_cursor[0], ctr, gstack, stack);
            __syncthreads();
            uint32_t *next = nullptr, *shadow = nullptr;
            uint32_t n_next = 0, n_shadow = 0;
            rp_shade_body<VARIANT, false, LIGHTS, TEX, true, TABLE, true>(sc, f, ps, sq, s_cur, n, nullptr, nullptr, nullptr, ctr, next, n_next, shadow, n_shadow,
                                                                                  &lds);
            __syncthreads();
            rp_connect_body<false, ALPHA, SINGLE, true, 0, true>(sc, f, ps, sq, shadow, n_shadow, &s_cursor[1], ctr, gstack, stack);
            __syncthreads();
            if (threadIdx.x < n_next) s_cur[threadIdx.x] = next[threadIdx.x]; // survivors: at most n <= RP_TAIL_CHUNK
            n = n_next;
        }
    }
}

// rp_kernarg (above) reads RpScene / RpFrame at the offsets they have as the FIRST TWO by-value arguments of a kernel: both kernels that run
// rp_shade_body must start their argument lists that way.
template <class F>
struct rp_args_start_with_scene_and_frame : std::false_type {};
template <class... Rest>
struct rp_args_start_with_scene_and_frame<void (*)(RpScene, RpFrame, Rest...)> : std::true_type {};
static_assert(rp_args_start_with_scene_and_frame<decltype(&rp_k_shade<RPTR_VARIANT_SIMPLE, true, false, false, false>)>::value &&
                  rp_args_start_with_scene_and_frame<decltype(&rp_k_tail<RPTR_VARIANT_SIMPLE, false, false, false, true, false>)>::value,
              "rp_k_shade / rp_k_tail: (RpScene, RpFrame, ...) must come first (kernels.h rp_kernarg)");
