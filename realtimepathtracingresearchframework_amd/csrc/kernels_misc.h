// kernels_misc.h -- the kernels around the path stages: resolve (accumulate.glsl +
// process_samples.comp), the ray-query kernel (rt_intersect.comp) and the refit of dynamic meshes. Included by rptr_hip.hip only
// (non-template kernels: one definition); the path stages themselves are templates in kernels.h, instantiated in k_*.hip.
#pragma once
#include "kernels.h"

// The queue of the first bounce is never stored: its entry i IS path id i (sample slot after sample slot, inside a slot the 8x8
// tiles row by row: 64 consecutive entries = one tile = one wave of camera rays). Ids of the tile padding beyond the right / bottom
// edge of a frame whose size is not a multiple of 8 name no pixel sample: rp_primary_ray returns false for them, the first extend
// gives them an empty interval (nothing is traversed), the first shade skips them.
// (Rounds 1-2 had a separate regrouping pass here -- rp_k_sort_count / _scan / _scatter: a counting sort of the extended paths by
// (material group, hit cell), three launches per bounce. It made the following shade launch up to 1.8x faster and cost more than it saved
// on every configuration; what north_star asks of it now lives inside the shade kernel's own LDS compaction, kernels.h rp_shade_body,
// RPTR_REGROUP.)

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
RP_DEV float4 rp_display_color(const RpFrame &f, float4 o, int pixel, int frame = 0) {
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
        {
            const float *cp = f.per_frame_cams != 0 ? f.cams[min(frame, RP_BATCH_CAMS - 1)].pos : f.cam_pos;
            o = make_float4((o.x - cp[0]) * 0.1f + 0.5f, (o.y - cp[1]) * 0.1f + 0.5f, (o.z - cp[2]) * 0.1f + 0.5f, o.w);
        }
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
                // REPROJECTION_MODE_DISCARD_HISTORY (process_samples.comp:116-131): the history of EARLIER frames is not folded in, a frame shows
                // all of its own samples -- the index inside the frame, which for a frame the backend splits into several internal launches
                // (spp > max_batch_spp) continues where the previous launch stopped (sample_base - frame_id samples of this frame came before)
                const uint32_t in_frame = (f.batch_frames > 1 ? 0u : f.sample_base - f.frame_id) + uint32_t(j);
                const uint32_t sample_index = f.rp.reprojection_mode == 1 ? in_frame : rp_slot_frame(f, uint32_t(s)).sample_index;
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
                o = rp_display_color(f, o, i, k);
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
            const int *tr = reinterpret_cast<const int *>(sc.tris + h.tri); // prim, geom: words 9 and 10 of the triangle record
            r = make_float4(h.u, h.v, __int_as_float(geometry_base + tr[10]), __int_as_float(tr[9]));
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
// shade: the triangles' shading records (dshade.h RpShadeTri) get the same new vertices
__global__ __launch_bounds__(256) void rp_k_refit_tris(RptrBvhTri *tris, float *tri_box, RpShadeTri *shade, uint32_t begin, uint32_t count,
                                                       const float *const *geom_dyn) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        RptrBvhTri t = tris[begin + i];
        const float *p = geom_dyn[t.geom] + 9ull * t.prim;
        float *b = tri_box + 6ull * (begin + i);
        float *sp = shade[begin + i].pos;
        for (int k = 0; k < 9; ++k) sp[k] = p[k];
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
// The shading records of the BVH triangles [begin, begin + count) (dshade.h RpShadeTri): after set_scene built or uploaded the triangles,
// and after a device-side rebuild reordered a mesh's. geometry_base >= 0: the triangles of one mesh, records of the parameterized mesh whose
// geometry records start there; < 0: a flattened scene, every triangle names its own instance record.
__global__ __launch_bounds__(256) void rp_k_build_shade_tris(RpScene sc, RpShadeTri *shade, uint32_t begin, uint32_t count, int geometry_base) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < count; i += gridDim.x * blockDim.x) {
        const RptrBvhTri t = sc.tris[begin + i];
        const int gb = geometry_base >= 0 ? geometry_base : sc.insts[RPTR_BVH_TRI_INSTANCE(t.flags)].geometry_base;
        shade[begin + i] = rp_make_shade_tri(sc.geoms[gb + (int)t.geom], t.prim);
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
