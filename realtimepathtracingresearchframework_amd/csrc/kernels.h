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

struct RpPathState {
    float4 *ray_o;   // origin.xyz, t_min
    float4 *ray_d;   // dir.xyz, t_max
    float4 *thr;     // throughput.xyz, prev_bounce_pdf
    float4 *illum;   // illum.xyz, bits(bounce)
    float2 *rng_tt;  // bits(rng state), total_t
    float4 *hit_tuv; // t, u, v, bits(prim)
    int2 *hit_ids;   // inst_idx, geom
};
// shadow rays live at the slot of their path (at most one per path and bounce);
// the shadow queue itself only carries path ids
struct RpShadowRays {
    float4 *o;       // origin.xyz, t_min
    float4 *d;       // dir.xyz, t_max
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

#define RP_SORT_MAX_KEYS 16384 // bins of the regrouping pass (64 KiB of LDS per block)
#define RP_SORT_BLOCKS 512
#define RP_SORT_MIN_N 32768u   // below this many paths the regrouping pass is skipped
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
RP_DEV bool rp_primary_ray_ex(const RpFrame &f, uint32_t p, uint32_t &rng, V3 &dir, int &lx, int &ly, uint32_t &sslot) {
    sslot = rp_div(p, f.div_npix_padded);
    const uint32_t slot = p - sslot * uint32_t(f.npix_padded);
    lx = ly = 0;
    if (!rp_slot_to_local(f, slot, lx, ly)) return false;
    const int gy = rp_local_row_to_global(f, ly);
    if (gy >= f.height) return false;
    const RpSlotFrame sf = rp_slot_frame(f, sslot);
    rng = rp_rng_seed(sf.sample_index, sf.frame_offset, uint32_t(lx), uint32_t(gy), uint32_t(f.width));
    V2 point = v2(float(lx) + 0.5f, float(gy) + 0.5f);
    if (f.rp.enable_raster_taa == 0) point = point + (rp_rand2(rng) - v2(0.5f, 0.5f));
    point = v2(point.x / float(f.width), point.y / float(f.height));
    dir = norm3(point.x * ld3(f.cam_du) + point.y * ld3(f.cam_dv) + ld3(f.cam_dir_top_left));
    return true;
}
RP_DEV bool rp_primary_ray(const RpFrame &f, uint32_t p, uint32_t &rng, V3 &dir) {
    int lx, ly;
    uint32_t sslot;
    return rp_primary_ray_ex(f, p, rng, dir, lx, ly, sslot);
}

// The queue of the first bounce is never stored: its entry i IS path id i (sample slot after sample slot, inside a slot the 8x8
// tiles row by row: 64 consecutive entries = one tile = one wave of camera rays). Ids of the tile padding beyond the right / bottom
// edge of a frame whose size is not a multiple of 8 name no pixel sample: rp_primary_ray returns false for them, the first extend
// gives them an empty interval (nothing is traversed), the first shade skips them.
// The same list in memory, for the opt-in regrouping pass (its kernels read a queue array):
__global__ __launch_bounds__(256) void rp_k_first_queue(uint32_t *queue, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) queue[i] = i;
}

// ------------------------------------------------------------------ extend (closest hit), persistent waves
// FIRST: bounce 0, the rays are the camera rays (computed, not loaded)
// ALPHA: the scene has alpha-tested materials. The test of a candidate may draw from the path's generator
// (pt_megakernel.glsl:354-358), so the lane carries it through the traversal and hands it back in the path state.
// SINGLE: the scene has one instance record; queries start inside it (dtraverse.h).
// LOCAL: `queue` / `cursor` are a block-local list and its cursor in LDS (rp_k_tail).
template <bool COUNT, bool FIRST, bool ALPHA, bool SINGLE, bool LOCAL>
RP_DEV void rp_extend_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const uint32_t *queue, uint32_t n, uint32_t *cursor, RpCounters *ctr,
                           int *gstack) {
    uint32_t n_nodes = 0, n_tris = 0;
    uint32_t lane_rng = 0, lane_rng_in = 0; // ALPHA only
    uint32_t lane_p = 0; // the path whose ray this lane traces
    auto load = [&](uint32_t i, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool {
        const uint32_t p = (FIRST && !queue) ? i : queue[i]; // FIRST: the first queue is the identity (NULL) unless the caller stored it
        lane_p = p;
        if (FIRST) {
            uint32_t rng = 0;
            rd = v3(0.0f, 0.0f, 1.0f);
            if (!rp_primary_ray(f, p, rng, rd)) { // tile padding: no such pixel sample. A miss is recorded (the regrouping pass reads it)
                ps.hit_ids[p] = make_int2(-1, -1);
                return false;
            }
            ro = ld3(f.cam_pos);
            tmin = 0.0f;
            tmax = 2.e32f;
            if (ALPHA) lane_rng = rng;
        } else {
            const float4 o = ps.ray_o[p], d = ps.ray_d[p];
            ro = xyz(o);
            rd = xyz(d);
            tmin = o.w;
            tmax = d.w;
            if (ALPHA) lane_rng = lane_rng_in = __float_as_uint(ps.rng_tt[p].x);
        }
        return true;
    };
    auto done = [&](uint32_t, const RpHitRec &h) {
        const uint32_t p = lane_p;
        ps.hit_tuv[p] = make_float4(h.t, h.u, h.v, __int_as_float(h.prim));
        ps.hit_ids[p] = make_int2(h.inst_idx, h.geom);
        if (ALPHA) {
            if (FIRST)
                ps.rng_tt[p] = make_float2(__uint_as_float(lane_rng), 0.0f); // the first shade takes it from here (f.alpha_test)
            else if (lane_rng != lane_rng_in)
                reinterpret_cast<float *>(ps.rng_tt + p)[0] = __uint_as_float(lane_rng);
        }
    };
    auto alpha = [&](uint32_t, int inst_idx, int, int geom, int prim, float u, float v) -> bool {
        return rp_alpha_rejects(sc, inst_idx, geom, prim, u, v, lane_rng);
    };
    rp_wave_trace<false, COUNT, (FIRST ? RP_NODE_MIN_FIRST : RP_NODE_MIN), (FIRST ? RP_REFILL_MIN_FIRST : RP_REFILL_MIN), ALPHA, SINGLE, LOCAL>(
        sc, n, cursor, gstack, load, done, alpha, n_nodes, n_tris);
    if (COUNT) {
        n_nodes = rp_wave_sum_u32(n_nodes);
        n_tris = rp_wave_sum_u32(n_tris);
        if (rp_lane_id() == 0) {
            atomicAdd(&ctr->nodes, (unsigned long long)n_nodes);
            atomicAdd(&ctr->tris, (unsigned long long)n_tris);
        }
    }
}
template <bool COUNT, bool FIRST, bool ALPHA, bool SINGLE>
__global__ RP_TRAVERSE_BOUNDS void rp_k_extend(RpScene sc, RpFrame f, RpPathState ps, const uint32_t *queue, RpBounceCounters *bc, RpCounters *ctr,
                                               int *gstack) {
    rp_extend_body<COUNT, FIRST, ALPHA, SINGLE, false>(sc, f, ps, queue, bc->queue_count, &bc->cursor_extend, ctr, gstack);
}

// ------------------------------------------------------------------ connect (shadow rays), persistent waves
// ALPHA: shadow rays test alpha-tested candidates with a generator seeded per candidate from (primitive ^ frame_id,
// instance ^ frame_offset, pixel), pt_megakernel.glsl:251-262 -- independent of the order in which candidates turn up.
// ids: the compacted path ids of the shadow rays (sq.ids, or the tail kernel's block-local list: LOCAL)
template <bool COUNT, bool ALPHA, bool SINGLE, bool LOCAL>
RP_DEV void rp_connect_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq, const uint32_t *ids, uint32_t n, uint32_t *cursor,
                            RpCounters *ctr, int *gstack) {
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
        const float4 o = sq.o[p], d = sq.d[p];
        ro = xyz(o);
        rd = xyz(d);
        tmin = o.w;
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
    rp_wave_trace<true, COUNT, RP_NODE_MIN_ANY, RP_REFILL_MIN_ANY, ALPHA, SINGLE, LOCAL>(sc, n, cursor, gstack, load, done, alpha, n_nodes, n_tris);
    if (COUNT) {
        n_nodes = rp_wave_sum_u32(n_nodes);
        n_tris = rp_wave_sum_u32(n_tris);
        if (rp_lane_id() == 0) {
            atomicAdd(&ctr->nodes_shadow, (unsigned long long)n_nodes);
            atomicAdd(&ctr->tris_shadow, (unsigned long long)n_tris);
        }
    }
}
template <bool COUNT, bool ALPHA, bool SINGLE>
__global__ RP_TRAVERSE_BOUNDS void rp_k_connect(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, RpBounceCounters *bc, RpCounters *ctr, int *gstack) {
    rp_connect_body<COUNT, ALPHA, SINGLE, false>(sc, f, ps, sq, sq.ids, bc->shadow_count, &bc->cursor_connect, ctr, gstack);
}

// ------------------------------------------------------------------ sort by material and hit cell
// Regroups the paths that were just extended so that a wave shades one material and
// -- just as important on this machine -- neighbouring hit points: the shadow rays
// and continuation rays it emits then start close together and share BVH nodes,
// which is what the L1 path (64 B/clk/CU, one distinct line per clock) rewards.
//   key 0                     = miss
//   1 + group*cells + cell    = hit; group = material id % groups, cell = position
//                               of the hit in a grid over the scene bounds
// One counting-sort pass with up to RP_SORT_MAX_KEYS bins: block-local LDS
// histograms (wave ballot aggregation for the dominant keys, ds_add for the rest),
// one global add per non-empty (block, bin), single-block scan, and a scatter that
// reserves a contiguous range per (block, bin).
RP_DEV uint32_t rp_sort_key(const RpScene &sc, const RpFrame &f, const RpPathState &ps, uint32_t p) {
    const int2 ids = ps.hit_ids[p];
    if (ids.x < 0) return 0u;
    const float4 hit = ps.hit_tuv[p];
    const int prim = __float_as_int(hit.w);
    const int geometry_base = reinterpret_cast<const int *>(sc.insts + ids.x)[13]; // RptrBvhInstance::geometry_base
    const RpGeomRecord &g = sc.geoms[geometry_base + ids.y];
    const int mid = rp_hit_material_id(g, uint32_t(prim));
    const float4 o = ps.ray_o[p], d = ps.ray_d[p];
    const float px = o.x + hit.x * d.x, py = o.y + hit.x * d.y, pz = o.z + hit.x * d.z;
    const int cx = min(max(int((px - f.sort_lo[0]) * f.sort_scale[0]), 0), (1 << f.sort_bits[0]) - 1);
    const int cy = min(max(int((py - f.sort_lo[1]) * f.sort_scale[1]), 0), (1 << f.sort_bits[1]) - 1);
    const int cz = min(max(int((pz - f.sort_lo[2]) * f.sort_scale[2]), 0), (1 << f.sort_bits[2]) - 1);
    const uint32_t cell = (uint32_t(cx) << (f.sort_bits[1] + f.sort_bits[2])) | (uint32_t(cy) << f.sort_bits[2]) | uint32_t(cz);
    const uint32_t group = uint32_t(mid) % uint32_t(f.sort_groups);
    return 1u + group * uint32_t(f.sort_cells) + cell;
}
RP_DEV void rp_sort_slice(uint32_t n, uint32_t &begin, uint32_t &end) {
    uint32_t per = (n + RP_SORT_BLOCKS - 1) / RP_SORT_BLOCKS;
    per = (per + 255u) & ~255u;
    begin = min(n, blockIdx.x * per);
    end = min(n, begin + per);
}
// table[key] += 1 for every valid lane; returns the previous value seen by the lane (its slot).
// The two most common keys of the wave are handled with ballot + one LDS add each.
RP_DEV uint32_t rp_lds_take(uint32_t *table, uint32_t key, bool valid) {
    const uint32_t lane = rp_lane_id();
    uint32_t pos = 0;
    unsigned long long todo = __ballot(valid);
#pragma unroll 1
    for (int it = 0; it < 2 && todo; ++it) {
        const int leader = __ffsll((long long)todo) - 1;
        const uint32_t k = __shfl(key, leader);
        const unsigned long long same = __ballot(valid && key == k);
        uint32_t b = 0;
        if (int(lane) == leader) b = atomicAdd(&table[k], (uint32_t)__popcll(same));
        b = __shfl(b, leader);
        if (valid && key == k) pos = b + (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) pos = atomicAdd(&table[key], 1u);
    return pos;
}
__global__ __launch_bounds__(256) void rp_k_sort_count(RpScene sc, RpFrame f, RpPathState ps, const uint32_t *queue, const uint32_t *count_ptr,
                                                       uint32_t *keys, uint32_t *hist) {
    __shared__ uint32_t lh[RP_SORT_MAX_KEYS];
    const uint32_t n = *count_ptr;
    if (n < RP_SORT_MIN_N) return;
    const int num_keys = f.sort_num_keys;
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    uint32_t begin, end;
    rp_sort_slice(n, begin, end);
    for (uint32_t i = begin + threadIdx.x; i < ((end + 255u) & ~255u) && begin < end; i += 256) {
        const bool valid = i < end;
        uint32_t key = 0;
        if (valid) {
            key = rp_sort_key(sc, f, ps, queue[i]);
            keys[i] = key;
        }
        (void)rp_lds_take(lh, key, valid);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x)
        if (lh[k]) atomicAdd(&hist[k], lh[k]);
}
// single block: exclusive scan of hist -> base; clears hist and the scatter cursors for the next bounce
__global__ __launch_bounds__(1024) void rp_k_sort_scan(uint32_t *hist, uint32_t *base, uint32_t *cursor, int num_keys) {
    __shared__ uint32_t partial[1024];
    const uint32_t total = uint32_t(num_keys);
    const uint32_t per = (total + 1023u) / 1024u;
    const uint32_t b = threadIdx.x * per, e = min(total, b + per);
    uint32_t sum = 0;
    for (uint32_t i = b; i < e; ++i) sum += hist[i];
    partial[threadIdx.x] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        uint32_t v = int(threadIdx.x) >= off ? partial[threadIdx.x - off] : 0u;
        __syncthreads();
        partial[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = partial[threadIdx.x] - sum;
    for (uint32_t i = b; i < e; ++i) {
        const uint32_t c = hist[i];
        base[i] = run;
        hist[i] = 0;
        cursor[i] = 0;
        run += c;
    }
}
__global__ __launch_bounds__(256) void rp_k_sort_scatter(RpFrame f, const uint32_t *queue, const uint32_t *count_ptr, const uint32_t *keys,
                                                         const uint32_t *base, uint32_t *cursor, uint32_t *order) {
    __shared__ uint32_t lh[RP_SORT_MAX_KEYS];
    const uint32_t n = *count_ptr;
    uint32_t begin, end;
    rp_sort_slice(n, begin, end);
    if (n < RP_SORT_MIN_N) { // too few paths for regrouping to pay: keep the queue order
        for (uint32_t i = begin + threadIdx.x; i < end; i += 256) order[i] = queue[i];
        return;
    }
    const int num_keys = f.sort_num_keys;
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    // pass A: this block's histogram of its slice
    for (uint32_t i = begin + threadIdx.x; i < ((end + 255u) & ~255u) && begin < end; i += 256) {
        const bool valid = i < end;
        (void)rp_lds_take(lh, valid ? keys[i] : 0u, valid);
    }
    __syncthreads();
    // reserve one contiguous output range per non-empty bin of this block
    for (int k = threadIdx.x; k < num_keys; k += blockDim.x) {
        const uint32_t c = lh[k];
        if (c) lh[k] = base[k] + atomicAdd(&cursor[k], c);
    }
    __syncthreads();
    // pass B: scatter
    for (uint32_t i = begin + threadIdx.x; i < ((end + 255u) & ~255u) && begin < end; i += 256) {
        const bool valid = i < end;
        const uint32_t pos = rp_lds_take(lh, valid ? keys[i] : 0u, valid);
        if (valid) order[pos] = queue[i];
    }
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
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX, bool LOCAL>
RP_DEV void rp_shade_body(const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq, const uint32_t *order, const uint32_t n,
                          uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count, RpCounters *ctr, uint32_t *&local_next, uint32_t &n_next,
                          uint32_t *&local_shadow, uint32_t &n_shadow) {
    __shared__ uint32_t s_next[RP_CHUNK], s_shadow[RP_CHUNK];
    __shared__ uint32_t s_nn, s_ns, s_base;
    __shared__ uint32_t s_stat[3];
    __shared__ uint32_t s_list[RP_CHUNK]; // path ids of the chunk, hits first
    __shared__ uint32_t s_nhit, s_nmiss;
    // per wave: the tri-light requests of its lanes (hit point, normal, bin) and the contributions of their bins
    __shared__ float s_ris_req[LIGHTS ? (256 / 64) * 64 * 8 : 1];
    __shared__ float s_ris_contrib[LIGHTS ? (256 / 64) * 64 * RPTR_BINNED_LIGHTS_BIN_MAX_SIZE : 1];
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
                pp = (FIRST && !order) ? i : order[i];
                is_hit = ps.hit_ids[pp].x >= 0;
            }
            const uint32_t ah = rp_wave_append(&s_nhit, valid && is_hit);
            if (valid && is_hit) s_list[ah] = pp;
            const uint32_t am = rp_wave_append(&s_nmiss, valid && !is_hit);
            if (valid && !is_hit) s_list[RP_CHUNK - 1 - am] = pp;
        }
        __syncthreads();
        const uint32_t chunk_hits = s_nhit, chunk_n = s_nhit + s_nmiss;
#pragma unroll 1
        for (uint32_t kk = 0; kk < RP_CHUNK / 256; ++kk) {
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
            uint32_t rng = 0;
            float total_t = 0.f, prev_bounce_pdf = 0.f, geometry_scale = 0.f;
            V3 ray_origin = v3s(0.f), ray_dir = v3s(0.f), throughput = v3s(0.f), illum = v3s(0.f), scatter_throughput = v3s(0.f);
            V3 ip_p = v3s(0.f), gn = v3s(0.f), nn = v3s(0.f), w_o = v3s(0.f), v_x = v3s(0.f), v_y = v3s(0.f);
            int bounce = 0;
            int aov_px = -1; // FIRST: local pixel whose AOVs this path writes
            RpMaterial mat;
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
            if (present) {
                p = il < chunk_hits ? s_list[il] : s_list[RP_CHUNK - 1 - (il - chunk_hits)];
                // bounce 0: the camera ray again; ids of tile padding name no pixel sample (the first queue is the identity)
                if (FIRST) present = rp_primary_ray_ex(f, p, rng, ray_dir, first_lx, first_ly, first_sslot);
            }
            if (present) {
                my_closest++;
                if (FIRST) { // init_shading_sample_state (shading_interface.glsl:20-22)
                    if (f.alpha_test) rng = __float_as_uint(ps.rng_tt[p].x); // alpha tests of the first extend may have drawn from it
                    if (f.aov_albedo_roughness) { // the first sample of the (last) frame (of the batch) writes the AOVs
                        const RpSlotFrame sf = rp_slot_frame(f, first_sslot);
                        if (sf.sample_index == sf.frame_id && int(sf.frame) == f.batch_frames - 1) aov_px = first_ly * f.width + first_lx;
                    }
                    ray_origin = ld3(f.cam_pos);
                    throughput = v3s(1.0f);
                    illum = v3s(0.0f);
                    prev_bounce_pdf = 2.e16f;
                    total_t = 0.0f;
                    bounce = 0;
                } else {
                    const float4 ro4 = ps.ray_o[p], rd4 = ps.ray_d[p];
                    const float4 thr4 = ps.thr[p];
                    const float4 il4 = ps.illum[p];
                    const float2 rt = ps.rng_tt[p];
                    rng = __float_as_uint(rt.x);
                    total_t = rt.y;
                    ray_origin = xyz(ro4);
                    ray_dir = xyz(rd4);
                    throughput = xyz(thr4);
                    illum = xyz(il4);
                    prev_bounce_pdf = thr4.w;
                    bounce = __float_as_int(il4.w);
                }
                const float4 hit4 = ps.hit_tuv[p];
                const int2 ids = ps.hit_ids[p];
                if (ids.x < 0) {
                    // miss: pt_megakernel.glsl:480-489
                    illum = illum + throughput * rp_compute_sky_illum(f, ray_dir, prev_bounce_pdf);
                    ps.illum[p] = f4(illum, __int_as_float(bounce));
                    if (FIRST && aov_px >= 0) { // pt_megakernel.glsl:482-487
                        rp_store_geometry_aovs(f, aov_px, v3s(0.0f), v3s(2.e32f));
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
                    // :585,605
                    total_t += hit.dist;
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
                        const float4 tx = rp_texture_lod0(sc, mp.normal_map, hit.uv);
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
                    rp_unpack_material<VARIANT, TEX>(sc, mat, emit, mp, hit.uv);
                    scatter_throughput = throughput;
                    if (FIRST && aov_px >= 0) { // pt_megakernel.glsl:670-673, shade_base_material.glsl:28-31
                        rp_store_geometry_aovs(f, aov_px, nn, ip_p);
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
                        dir_sample = rp_rand2(rng);
                        sel_sample = rp_rand2(rng);
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
                            has_shadow = true;
                            sq.o[p] = f4(ip_p, epsilon);
                            sq.d[p] = f4(light_dir, light_dist - epsilon);
                            sq.contrib[p] = f4(c, 0.0f);
                        } else
                            illum = illum + c; // visibility defaults to true
                    }
                }
            }
            // ---------------- C: continuation
            if (hit_lane) {
                if (!terminate && f.rp.glossy_only_mode != 0 && !(mat.roughness < 0.1f && mat.ior != 1.0f)) terminate = true;
                if (!terminate) {
                    const V2 lobe_sample = rp_rand2(rng);
                    const V2 dir_sample2 = rp_rand2(rng);
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
                        // pt_megakernel.glsl:703-709
                        ray_dir = w_i;
                        ray_origin = ip_p;
                        const float t_min = rp_geometry_scale_to_tmin(ray_origin, total_t);
                        // :713-730 Russian roulette
                        bool survive = true;
                        if (bounce >= f.rp.rr_path_depth) {
                            const float prefix_weight = fmaxf(throughput.x, fmaxf(throughput.y, throughput.z));
                            float rr_prob = prefix_weight;
                            const float rr_sample = rp_randf(rng);
                            rr_prob = (bounce > 6) ? fminf(0.95f, rr_prob) : fminf(1.0f, rr_prob);
                            if (rr_sample < rr_prob)
                                throughput = throughput / rr_prob;
                            else
                                survive = false;
                        }
                        if (survive) {
                            alive = true;
                            ps.ray_o[p] = f4(ray_origin, t_min);
                            ps.ray_d[p] = f4(ray_dir, 1e20f);
                            ps.thr[p] = f4(throughput, prev_bounce_pdf);
                            ps.rng_tt[p] = make_float2(__uint_as_float(rng), total_t);
                        }
                    }
                }
                ps.illum[p] = f4(illum, __int_as_float(bounce));
            }
            const uint32_t at = rp_wave_append(&s_nn, alive);
            if (alive) s_next[at] = p;
            const uint32_t sat = rp_wave_append(&s_ns, has_shadow);
            if (has_shadow) s_shadow[sat] = p;
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
template <int VARIANT, bool FIRST, bool LIGHTS, bool TEX>
__global__ __launch_bounds__(256, RP_SHADE_WAVES) void rp_k_shade(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, const uint32_t *order,
                                                  const uint32_t *count_ptr, uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count,
                                                  RpCounters *ctr) {
    uint32_t *ln = nullptr, *ls = nullptr;
    uint32_t nn = 0, ns = 0;
    rp_shade_body<VARIANT, FIRST, LIGHTS, TEX, false>(sc, f, ps, sq, order, *count_ptr, next_queue, next_count, shadow_count, ctr, ln, nn, ls, ns);
}

// ------------------------------------------------------------------ tail: the late bounces of a frame in ONE launch
// From some bounce on a frame's queues hold a few thousand paths, and what a bounce then costs is its three launches
// (a command-processor packet each, ~14 us when several frames are in flight: profiles/r01_notes.md), not its rays. Paths are
// independent, so the rest of the frame needs no grid-wide step: a block takes RP_TAIL_CHUNK paths of the bounce's queue and runs
// them to the end -- extend, shade, connect per bounce on block-local lists in LDS, the same device code as the stand-alone
// kernels (results are bit-identical, tests/test_gpu_parity.py) -- before it takes the next chunk.
#define RP_TAIL_CHUNK 256
template <int VARIANT, bool LIGHTS, bool TEX, bool ALPHA, bool SINGLE>
__global__ __launch_bounds__(256, 1) void rp_k_tail(RpScene sc, RpFrame f, RpPathState ps, RpShadowRays sq, const uint32_t *queue, RpCounters *ctr,
                                                    int first_bounce, int *gstack) {
    __shared__ uint32_t s_cur[RP_TAIL_CHUNK];
    __shared__ uint32_t s_cursor[2];
    const uint32_t n_total = ctr->bounce[first_bounce].queue_count;
    const uint32_t nchunks = (n_total + RP_TAIL_CHUNK - 1) / RP_TAIL_CHUNK;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        uint32_t n = min((uint32_t)RP_TAIL_CHUNK, n_total - chunk * RP_TAIL_CHUNK);
        __syncthreads(); // the previous chunk is done with s_cur
        if (threadIdx.x < n) s_cur[threadIdx.x] = queue[chunk * RP_TAIL_CHUNK + threadIdx.x];
        for (int b = first_bounce; b < f.rp.max_path_depth && n > 0; ++b) {
            if (threadIdx.x < 2) s_cursor[threadIdx.x] = 0;
            __syncthreads();
            rp_extend_body<false, false, ALPHA, SINGLE, true>(sc, f, ps, s_cur, n, &s_cursor[0], ctr, gstack);
            __syncthreads();
            uint32_t *next = nullptr, *shadow = nullptr;
            uint32_t n_next = 0, n_shadow = 0;
            rp_shade_body<VARIANT, false, LIGHTS, TEX, true>(sc, f, ps, sq, s_cur, n, nullptr, nullptr, nullptr, ctr, next, n_next, shadow, n_shadow);
            __syncthreads();
            rp_connect_body<false, ALPHA, SINGLE, true>(sc, f, ps, sq, shadow, n_shadow, &s_cursor[1], ctr, gstack);
            __syncthreads();
            if (threadIdx.x < n_next) s_cur[threadIdx.x] = next[threadIdx.x]; // survivors: at most n <= RP_TAIL_CHUNK
            n = n_next;
        }
    }
}

// the query kernel (rp_k_trace) borrows a pool cursor: reset it
__global__ void rp_k_reset_u32(uint32_t *p) { *p = 0; }

// ------------------------------------------------------------------ resolve
// accumulate.glsl:68-73 (store this sample) + process_samples.comp:116-132 (running mean into the
// history) + :143-198 (exposure, early tone mapping, AOV views, sRGB, RGBA8). One thread per local pixel, samples folded in order.
// out_accum / out_fb (frames in flight, else NULL): a second copy of what this frame leaves in accum / fb
RP_DEV float4 rp_half4_to_float4(uint2 h) {
    return make_float4((float)__builtin_bit_cast(_Float16, (uint16_t)(h.x & 0xFFFFu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(h.x >> 16)),
                       (float)__builtin_bit_cast(_Float16, (uint16_t)(h.y & 0xFFFFu)), (float)__builtin_bit_cast(_Float16, (uint16_t)(h.y >> 16)));
}
// rendering/postprocess/tonemapping_utils.glsl:9-33 (modes: postprocess/tonemapping.h)
RP_DEV V3 rp_tonemap(int mode, V3 c) {
    if (mode == 2) // FAST_TONE_MAPPING
        return c / (v3s(1.0f) + c);
    if (mode == 1) { // NEUTRAL_TONE_MAPPING
        const float luminance_level = fmaxf(fmaxf(c.x, c.y), fmaxf(c.z, 1.0f));
        return c * (mixf(0.1f * log2f(luminance_level), 1.0f, 0.8f) / luminance_level);
    }
    return c; // NO_TONE_MAPPING
}
// process_samples.comp:143-190: what the RGBA8 frame buffer shows for the resolved pixel `acc` (alpha already clamped)
RP_DEV float4 rp_display_color(const RpFrame &f, float4 o, int pixel) {
    const int ch = f.rp.output_channel;
    if (ch == 0) { // OUTPUT_CHANNEL_COLOR
        const float e = exp2f(f.rp.exposure);
        V3 c = v3(o.x * e, o.y * e, o.z * e);
        if (f.rp.early_tone_mapping_mode >= 0) c = rp_tonemap(f.rp.early_tone_mapping_mode, c);
        o = f4(c, o.w);
    } else if (f.aov_albedo_roughness) { // ENABLE_AOV_BUFFERS: the views of the AOV images
        if (ch == 1) {
            o = rp_half4_to_float4(f.aov_albedo_roughness[pixel]);
            if (f.rp.output_moment != 0) o = make_float4(o.w, o.w, o.w, o.w);
        } else if (ch == 2) {
            o = rp_half4_to_float4(f.aov_normal_depth[pixel]);
            if (f.rp.output_moment != 0)
                o = make_float4(o.w * 0.05f, o.w * 0.05f, o.w * 0.05f, o.w);
            else
                o = make_float4(o.x * 0.5f + 0.5f, o.y * 0.5f + 0.5f, o.z * 0.5f + 0.5f, o.w);
        } else if (ch == 3) {
            const float4 mj = rp_half4_to_float4(f.aov_motion_jitter[pixel]);
            if (f.rp.output_moment == 0)
                o = make_float4(fabsf(10.0f * mj.x), fabsf(10.0f * mj.y), 0.0f, 1.0f);
            else { // jitter back to pixel units (process_samples.comp:171-176)
                const float jx = (mj.z + 1.0f / float(f.width)) * (float(f.width) / 2.0f), jy = (mj.w + 1.0f / float(f.height)) * (float(f.height) / 2.0f);
                o = make_float4(jx * 0.5f + 0.5f, jy * 0.5f + 0.5f, 0.0f, 1.0f);
            }
        }
    } else { // without AOV images (RPTR_AOVS=0): the views of what the integrator accumulated (process_samples.comp:179-188)
        if (ch == 2) {
            if (f.rp.output_moment != 0) {
                const float l = len3(v3(o.x, o.y, o.z));
                o = make_float4(l, l, l, o.w);
            } else
                o = make_float4(o.x * 0.5f + 0.5f, o.y * 0.5f + 0.5f, o.z * 0.5f + 0.5f, o.w);
        } else if (ch == 3)
            o = make_float4((o.x - f.cam_pos[0]) * 0.1f + 0.5f, (o.y - f.cam_pos[1]) * 0.1f + 0.5f, (o.z - f.cam_pos[2]) * 0.1f + 0.5f, o.w);
    }
    return make_float4(rp_linear_to_srgb(o.x), rp_linear_to_srgb(o.y), rp_linear_to_srgb(o.z), o.w);
}
// out_accum / out_fb (frames in flight, else NULL): what each frame of the batch leaves in accum / fb, frame k at k * f.out_stride
__global__ __launch_bounds__(256) void rp_k_resolve(RpFrame f, RpPathState ps, float4 *accum, uchar4 *fb, float4 *out_accum, uchar4 *out_fb) {
    const int npix = f.width * f.local_rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += gridDim.x * blockDim.x) {
        const int ly = int(rp_div(uint32_t(i), f.div_width)), lx = i - ly * f.width;
        if (rp_local_row_to_global(f, ly) >= f.height) continue;
        const uint32_t slot = rp_local_to_slot(f, lx, ly);
        float4 acc = accum[i];
        uchar4 shown = fb[i];
        const int per_frame = f.batch_frames > 1 ? f.frame_spp : f.batch_spp;
        for (int k = 0; k < f.batch_frames; ++k) {
            for (int j = 0; j < per_frame; ++j) {
                const int s = k * per_frame + j;
                const float4 il = ps.illum[size_t(s) * size_t(f.npix_padded) + slot];
                const float4 c = make_float4(il.x, il.y, il.z, __float_as_int(il.w) == 0 ? 0.0f : 1.0f); // pt_megakernel.glsl:736
                const uint32_t sample_index = rp_slot_frame(f, uint32_t(s)).sample_index;
                if (sample_index == 0)
                    acc = c;
                else {
                    const float denom = float(int(sample_index) + 1);
                    acc.x += (c.x - acc.x) / denom;
                    acc.y += (c.y - acc.y) / denom;
                    acc.z += (c.z - acc.z) / denom;
                    acc.w += (c.w - acc.w) / denom;
                }
            }
            float4 o = acc;
            o.w = fminf(o.w, 1.0f);
            if (o.w >= 0.0f) {
                o = rp_display_color(f, o, i);
                shown = make_uchar4((unsigned char)(clamp1(o.x, 0.f, 1.f) * 255.0f + 0.5f), (unsigned char)(clamp1(o.y, 0.f, 1.f) * 255.0f + 0.5f),
                                    (unsigned char)(clamp1(o.z, 0.f, 1.f) * 255.0f + 0.5f), (unsigned char)(clamp1(o.w, 0.f, 1.f) * 255.0f + 0.5f));
            }
            if (out_accum) {
                out_accum[size_t(k) * f.out_stride + size_t(i)] = acc;
                out_fb[size_t(k) * f.out_stride + size_t(i)] = shown;
            }
        }
        accum[i] = acc;
        fb[i] = shown;
    }
}

// ------------------------------------------------------------------ RQ_CLOSEST, vulkan/rt_intersect.comp:31-68
// COUNT: also writes per-query visit counts (nodes, triangles) -- the diagnostic behind rptr_hip_trace_counted
// ANY (diagnostic only): occlusion query over (tmin_arr[i], t_max), result.x = 1 when anything is hit
template <bool COUNT, bool ANY, bool SINGLE>
__global__ RP_TRAVERSE_BOUNDS void rp_k_trace(RpScene sc, const RptrRenderRayQuery *queries, uint32_t n, float4 *results, uint32_t *cursor,
                                              int *gstack, uint2 *per_ray, const float *tmin_arr) {
    uint32_t nn = 0, nt = 0, nn_prev = 0, nt_prev = 0;
    auto load = [&](uint32_t i, V3 &ro, V3 &rd, float &tmin, float &tmax) -> bool {
        const float4 *qp = reinterpret_cast<const float4 *>(queries + i);
        const float4 q0 = qp[0], q1 = qp[1];
        ro = v3(q0.x, q0.y, q0.z);
        rd = v3(q1.x, q1.y, q1.z);
        tmin = tmin_arr ? tmin_arr[i] : RPTR_RAY_EPSILON * len3(ro); // rt_intersect.comp:40
        tmax = __float_as_int(q0.w) < 0 ? -1.0f : q1.w;  // mode < 0: skipped query, empty interval
        return true;
    };
    auto done = [&](uint32_t i, const RpHitRec &h) {
        if (COUNT && per_ray) { // the lane's counters run across its queries: report the difference
            per_ray[i] = make_uint2(nn - nn_prev, nt - nt_prev);
            nn_prev = nn;
            nt_prev = nt;
        }
        if (queries[i].mode_or_data < 0) return; // slot stays untouched (rt_intersect.comp:43-44)
        float4 r;
        if (ANY)
            r = make_float4(h.inst_idx < 0 ? 0.0f : 1.0f, 0.0f, 0.0f, 0.0f);
        else if (h.inst_idx < 0)
            r = make_float4(-1.0f, -1.0f, __int_as_float(-1), __int_as_float(-1));
        else {
            const int geometry_base = reinterpret_cast<const int *>(sc.insts + h.inst_idx)[13];
            r = make_float4(h.u, h.v, __int_as_float(geometry_base + h.geom), __int_as_float(h.prim));
        }
        results[i] = r;
    };
    // ray queries see opaque geometry
    rp_wave_trace<ANY, COUNT, (ANY ? RP_NODE_MIN_ANY : RP_NODE_MIN), (ANY ? RP_REFILL_MIN_ANY : RP_REFILL_MIN), false, SINGLE>(sc, n, cursor, gstack, load, done, RpNoAlpha(), nn, nt);
}

// ------------------------------------------------------------------ refit (dynamic meshes)
// Stands in for the driver's acceleration-structure UPDATE builds (vulkan/vulkanrt_utils.h:83-105,
// enqueue_refit): triangles are re-derived from the float vertex buffer, node boxes are recomputed
// bottom-up one height level per launch, instance bounds from the BLAS roots, then the TLAS levels.
// tri_box: bounds of the three VERTICES (what the builder bounds, bvh_build.cpp), kept for the node pass
__global__ __launch_bounds__(256) void rp_k_refit_tris(RptrBvhTri *tris, float *tri_box, uint32_t begin, uint32_t count,
                                                       const float *const *geom_dyn) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        RptrBvhTri t = tris[begin + i];
        const float *p = geom_dyn[t.geom] + 9ull * t.prim;
        float *b = tri_box + 6ull * (begin + i);
        for (int k = 0; k < 3; ++k) {
            t.v0[k] = p[k];
            t.e1[k] = p[3 + k] - p[k];
            t.e2[k] = p[6 + k] - p[k];
            b[k] = fminf(p[k], fminf(p[3 + k], p[6 + k]));
            b[3 + k] = fmaxf(p[k], fmaxf(p[3 + k], p[6 + k]));
        }
        tris[begin + i] = t;
    }
}
// one height level of nodes: child boxes from the triangle / instance bounds (leaves) or from the exact float
// bounds of the child nodes (node_box, written by the level below), then the shared encoder (bvh4.h)
RP_DEV void rp_refit_node(RptrBvh4Node *nodes, float *node_box, const float *tri_box, const float *inst_box, const uint32_t e) {
    const bool tlas = (e >> 31) != 0;
    const uint32_t ni = e & 0x7FFFFFFFu;
    int32_t child[4];
    RpBox4 b;
    for (int k = 0; k < 4; ++k) {
        const int32_t c = nodes[ni].child[k];
        child[k] = c;
        for (int a = 0; a < 3; ++a) {
            b.lo[k][a] = INFINITY;
            b.hi[k][a] = -INFINITY;
        }
        if (c == RPTR_BVH4_EMPTY) continue;
        if (c >= 0) {
            const float *nb = node_box + 6ull * c;
            for (int a = 0; a < 3; ++a) {
                b.lo[k][a] = nb[a];
                b.hi[k][a] = nb[3 + a];
            }
        } else {
            const int first = RPTR_BVH_LEAF_FIRST(c), count = RPTR_BVH_LEAF_COUNT(c);
            for (int j = 0; j < count; ++j) {
                const float *lb = (tlas ? inst_box : tri_box) + 6ull * (first + j);
                for (int a = 0; a < 3; ++a) {
                    b.lo[k][a] = fminf(b.lo[k][a], lb[a]);
                    b.hi[k][a] = fmaxf(b.hi[k][a], lb[3 + a]);
                }
            }
        }
    }
    RptrBvh4Node n;
    float *nb = node_box + 6ull * ni;
    rp_bvh4_encode(b, child, &n, nb, nb + 3);
    nodes[ni] = n;
}
RP_DEV void rp_refit_instance(const float *node_box, const RptrBvhInstance *insts, float *inst_box, uint32_t i) {
    const RptrBvhInstance &in = insts[i];
    const float *mb = node_box + 6ull * in.blas_root; // exact bounds of the mesh
    float lo[3], hi[3];
    for (int k = 0; k < 3; ++k) {
        lo[k] = INFINITY;
        hi[k] = -INFINITY;
    }
    const float *M = in.object_to_world;
    for (int c = 0; c < 8; ++c) {
        const float p[3] = {c & 1 ? mb[3] : mb[0], c & 2 ? mb[4] : mb[1], c & 4 ? mb[5] : mb[2]};
        for (int rr = 0; rr < 3; ++rr) {
            const float w = ((M[4 * rr] * p[0] + M[4 * rr + 1] * p[1]) + M[4 * rr + 2] * p[2]) + M[4 * rr + 3];
            lo[rr] = fminf(lo[rr], w);
            hi[rr] = fmaxf(hi[rr], w);
        }
    }
    for (int k = 0; k < 3; ++k) {
        inst_box[6 * i + k] = lo[k];
        inst_box[6 * i + 3 + k] = hi[k];
    }
}
__global__ __launch_bounds__(256) void rp_k_refit_nodes(RptrBvh4Node *nodes, float *node_box, const float *tri_box, const float *inst_box,
                                                        const uint32_t *list, uint32_t begin, uint32_t end) {
    for (uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
        rp_refit_node(nodes, node_box, tri_box, inst_box, list[i]);
}
__global__ __launch_bounds__(256) void rp_k_refit_instances(const float *node_box, const RptrBvhInstance *insts, float *inst_box, uint32_t n) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) rp_refit_instance(node_box, insts, inst_box, i);
}
// The small top of a refit in ONE launch of one block: the shallow bottom-level levels of one dynamic mesh (a level waits for the one
// below: a block barrier instead of a launch), the instance bounds, the top-level levels. `blas_levels` / `tlas_levels` hold [begin, end)
// pairs into their lists, in processing order. Same per-node arithmetic as the stand-alone kernels.
__global__ __launch_bounds__(1024) void rp_k_refit_top(RptrBvh4Node *nodes, float *node_box, const float *tri_box, float *inst_box, const uint32_t *blas_list,
                                                       const uint2 *blas_levels, int n_blas, const uint32_t *tlas_list, const uint2 *tlas_levels, int n_tlas,
                                                       const RptrBvhInstance *insts, uint32_t n_insts) {
    for (int l = 0; l < n_blas; ++l) {
        const uint2 lv = blas_levels[l];
        for (uint32_t i = lv.x + threadIdx.x; i < lv.y; i += blockDim.x) rp_refit_node(nodes, node_box, tri_box, inst_box, blas_list[i]);
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n_insts; i += blockDim.x) rp_refit_instance(node_box, insts, inst_box, i);
    __syncthreads();
    for (int l = 0; l < n_tlas; ++l) {
        const uint2 lv = tlas_levels[l];
        for (uint32_t i = lv.x + threadIdx.x; i < lv.y; i += blockDim.x) rp_refit_node(nodes, node_box, tri_box, inst_box, tlas_list[i]);
        __syncthreads();
    }
}
