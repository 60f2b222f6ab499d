// rptr_hip.hip -- C ABI of the MI355X wavefront path tracer (include/rptr_hip.h).
//
// Host side of the backend: owns device memory, builds the acceleration
// structures, sequences the wavefront stages on one HIP stream and does the
// frame bookkeeping of RenderVulkan::begin_frame/draw_frame/end_frame
// (vulkan/render_vulkan.cpp:1919-2178). There is no CPU rendering fallback: every
// compute entry point fails with RPTR_E_NO_DEVICE / RPTR_E_HIP when no GPU works.
#include "../../include/rptr_hip.h"

#include "bvh_build.h"
#include "kernels_misc.h"
#include "launch.h"
#include "lbvh.h"
#include "ploc.h"
#include <hip/hip_ext.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <map>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <type_traits>
#include <vector>

#include "host_state.h"
#include "host_bvh.inl"

extern "C" {

const char *rptr_hip_name(void) { return "HIP wavefront path tracer (gfx950)"; }

const char *rptr_hip_last_error(const rptr_hip_t *h) { return h ? h->last_error.c_str() : g_last_error.c_str(); }

int rptr_hip_abi_version(void) { return RPTR_HIP_ABI_VERSION; }

#ifndef RP_BUILD_ID
#define RP_BUILD_ID "unknown"
#endif
const char *rptr_hip_build_id(void) { return RP_BUILD_ID; }

int rptr_hip_bvh_build_info(rptr_hip_t *h, int32_t *out_device_built, float *out_build_ms, float *out_device_ms) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (out_device_built) *out_device_built = h->bvh_device_built ? 1 : 0;
    if (out_build_ms) *out_build_ms = (float)h->bvh_build_ms;
    if (out_device_ms) *out_device_ms = (float)h->bvh_device_ms;
    return RPTR_OK;
}

int rptr_hip_traversal_preset(rptr_hip_t *h, float *out_area_cost, int32_t *out_node_min, int32_t *out_refill_min) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (out_area_cost) *out_area_cost = (float)h->bvh_area_cost;
    if (out_node_min) *out_node_min = h->master.dscene.node_min;
    if (out_refill_min) *out_refill_min = h->master.dscene.refill_min;
    return RPTR_OK;
}

int rptr_hip_create(const RptrCreateInfo *info, rptr_hip_t **out) {
    if (!out) return fail(nullptr, RPTR_E_INVALID, "rptr_hip_create: out is NULL");
    *out = nullptr;
    if (info && info->abi_version != RPTR_HIP_ABI_VERSION)
        return fail(nullptr, RPTR_E_INVALID, "rptr_hip_create: RptrCreateInfo.abi_version is %d, this library implements version %d of include/rptr_hip.h "
                                             "(set abi_version = RPTR_HIP_ABI_VERSION; struct fields that used to be padding carry meaning now)",
                    info->abi_version, RPTR_HIP_ABI_VERSION);
    ensure_hw_queues(info ? info->frames_in_flight : 1, info && (info->flags & RPTR_CREATE_SET_HW_QUEUES) != 0u);
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return fail(nullptr, RPTR_E_NO_DEVICE, "no HIP device available (%s); this backend has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    rptr_hip *h = new rptr_hip();
    memset(&h->stats, 0, sizeof(h->stats));
    memset(&h->master.dscene, 0, sizeof(h->master.dscene));
    h->device = info ? info->device_ordinal : 0;
    h->rank = info ? info->rank : 0;
    h->world = info && info->world_size > 0 ? info->world_size : 1;
    h->stripe_rows = info && info->stripe_rows > 0 ? info->stripe_rows : 32;
    if (h->stripe_rows % 8 != 0) {
        delete h;
        return fail(nullptr, RPTR_E_INVALID, "stripe_rows must be a multiple of 8");
    }
    if (h->device < 0 || h->device >= n_dev || h->rank < 0 || h->rank >= h->world) {
        delete h;
        return fail(nullptr, RPTR_E_INVALID, "bad device ordinal %d (of %d) or rank %d/%d", h->device, n_dev, h->rank, h->world);
    }
    if (hipSetDevice(h->device) != hipSuccess) {
        delete h;
        return fail(nullptr, RPTR_E_NO_DEVICE, "hipSetDevice(%d) failed", info ? info->device_ordinal : 0);
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) {
        delete h;
        return fail(nullptr, RPTR_E_NO_DEVICE, "hipGetDeviceProperties failed");
    }
    h->num_cus = prop.multiProcessorCount;
    if (info && info->stream) {
        h->stream = (hipStream_t)info->stream;
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return fail(nullptr, RPTR_E_HIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    {
        int fif = info ? info->frames_in_flight : 1;
        if (const char *s = getenv("RPTR_FRAMES_IN_FLIGHT")) fif = atoi(s);
        fif = std::max(1, std::min(fif, 16));
        // connect(b) -- the shadow rays of bounce b -- on a side stream beside extend(b + 1): both only depend on shade(b). With ONE frame
        // context nothing else fills the ramp-down of a launch: a single frame gets 2-4 % shorter (C2 2.07 -> 2.04 ms, C3 6.52 -> 6.31, a 1/8
        // frame 0.83 -> 0.80, profiles/r03_notes.md). With frames in flight the extra stream only gets in the way of the other frames' launches
        // (+5 % pipelined): off there. Option "side_connect" = 0 | 1 overrides (the side streams are made by rptr_hip_initialize).
        h->ctx.resize((size_t)fif);
        h->opt = process_default_options();
        apply_option_env(h->opt);
        sync_options(h);
        for (FrameCtx &c : h->ctx) {
            memset(&c.ps, 0, sizeof(c.ps));
            memset(&c.sq, 0, sizeof(c.sq));
            memset(&c.earlier_batches, 0, sizeof(c.earlier_batches));
            if (fif == 1)
                c.stream = h->stream;
            else {
                if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
                    delete h;
                    return fail(nullptr, RPTR_E_HIP, "hipStreamCreate failed");
                }
                c.own_stream = true;
            }
            (void)hipEventCreate(&c.ev_begin);
            (void)hipEventCreate(&c.ev_end);
            (void)hipEventCreateWithFlags(&c.ev_dep, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&c.ev_resolved, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming);
            (void)hipEventCreateWithFlags(&c.ev_side, hipEventDisableTiming);
            if (hipHostMalloc((void **)&c.host_counters, sizeof(RpCounters), hipHostMallocDefault) != hipSuccess) {
                delete h;
                return fail(nullptr, RPTR_E_NOMEM, "hipHostMalloc failed");
            }
        }
    }
    // defaults of RenderParams / LightSamplingConfig (librender/render_params.glsl.h:123-155)
    memset(&h->params, 0, sizeof(h->params));
    h->params.batch_spp = 1;
    h->params.max_path_depth = RPTR_MAX_PATH_DEPTH;
    h->params.rr_path_depth = RPTR_DEFAULT_RR_PATH_DEPTH;
    h->params.focus_distance = 2.5f;
    h->params.pixel_radius = 1.0f;
    h->params.variance_radius = 4.0f;
    h->params.early_tone_mapping_mode = -1;
    h->params.spp_accumulation_window = 8;
    h->params.render_upscale_factor = 1;
    h->params.focal_length = 35.0f;
    h->lighting = RptrLightSamplingConfig{0.0f, 16, 15.0f, 0.0f};
    memset(&h->scene_params, 0, sizeof(h->scene_params));
    h->scene_params.sun_dir[1] = 1.0f;
    h->scene_params.sun_cos_angle = 0.99998933f;
    h->scene_params.sun_radiance[3] = 1.0f;
    h->scene_params.normal_z_scale = 1.0f;
    *out = h;
    return RPTR_OK;
}

void rptr_hip_destroy(rptr_hip_t *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (FrameCtx &c : h->ctx) (void)hipStreamSynchronize(c.stream);
    (void)hipStreamSynchronize(h->stream);
    comm_release(h);
    release_scene_copy_host(h->master);
    for (SceneCopy &sc : h->ctx_scene) release_scene_copy_host(sc);
    free_list(h->allocations);
    free_list(h->scene_allocs);
    for (FrameCtx &c : h->ctx) {
        for (hipEvent_t e : c.ev_pool) (void)hipEventDestroy(e);
        if (c.side) {
            (void)hipStreamSynchronize(c.side);
            (void)hipStreamDestroy(c.side);
        }
        for (hipEvent_t e : {c.ev_begin, c.ev_end, c.ev_dep, c.ev_resolved, c.ev_fork, c.ev_side, c.ev_gather})
            if (e) (void)hipEventDestroy(e);
        if (c.host_counters) (void)hipHostFree(c.host_counters);
        if (c.own_stream) (void)hipStreamDestroy(c.stream);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    if (h->rng_table) (void)hipFree(h->rng_table);
    if (h->rq_queries) (void)hipFree(h->rq_queries);
    if (h->rq_results) (void)hipFree(h->rq_results);
    delete h;
}

int rptr_hip_set_stream(rptr_hip_t *h, void *hip_stream) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    int rc = drain(h);
    if (rc) return rc;
    (void)hipStreamSynchronize(h->stream);
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    h->own_stream = false;
    if (hip_stream)
        h->stream = (hipStream_t)hip_stream;
    else {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        h->own_stream = true;
    }
    if (h->ctx.size() == 1) h->ctx[0].stream = h->stream;
    return RPTR_OK;
}

int rptr_hip_initialize(rptr_hip_t *h, int fb_width, int fb_height) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (fb_width <= 0 || fb_height <= 0) return fail(h, RPTR_E_INVALID, "bad framebuffer size %dx%d", fb_width, fb_height);
    HIP_TRY(h, hipSetDevice(h->device));
    {
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    comm_release(h); // its receive buffers and the assembled frame are frame-sized: a communicator is made again after a resize
    for (void *p : h->allocations) (void)hipFree(p);
    h->allocations.clear();
    h->bytes_frame = 0;
    h->bytes_allocated = h->bytes_scene;
    h->width = fb_width;
    h->height = fb_height;
    h->local_rows = local_row_count(fb_height, h->stripe_rows, h->rank, h->world);
    // (8 x 8 pixel tiles, numbered in blocks of RP_TILE_BLOCK x RP_TILE_BLOCK: dshade.h rp_slot_to_local)
    h->tiles_x = ((fb_width + 7) / 8 + RP_TILE_BLOCK - 1) / RP_TILE_BLOCK * RP_TILE_BLOCK;
    h->tiles_y = ((std::max(h->local_rows, 1) + 7) / 8 + RP_TILE_BLOCK - 1) / RP_TILE_BLOCK * RP_TILE_BLOCK;
    h->npix_padded = h->tiles_x * h->tiles_y * 64;
    // sample slots in flight: as many as fit a ~6 GiB path-state budget per frame context (288 GB of HBM), at most 16
    const size_t bytes_per_path = 16 * 5 + 8 + 2 * 16 + 5 * 4;
    // (options "path_budget_mb", "max_batch_spp", "max_batch_frames", "aovs", "side_connect", "blocks_per_cu" take effect here)
    sync_options(h);
    const size_t budget = (size_t)h->opt.v[OPT_PATH_BUDGET_MB] << 20;
    int mb = (int)std::min<size_t>(16, std::max<size_t>(1, budget / (bytes_per_path * (size_t)h->npix_padded)));
    if (h->opt.v[OPT_MAX_BATCH_SPP] > 0) mb = (int)h->opt.v[OPT_MAX_BATCH_SPP];
    h->max_batch_spp = mb;
    h->max_batch_frames = (int)h->opt.v[OPT_MAX_BATCH_FRAMES];
    h->aovs = h->opt.v[OPT_AOVS] != 0;
    // (auto: handles with one or two frame contexts have side streams; a frame uses its context's when no other frame of the handle is in
    // flight at its submission -- the synchronous loop of a host that holds the reference's two swap buffers, rptr_hip_render)
    h->side_connect = h->opt.v[OPT_SIDE_CONNECT] >= 0 ? (int)h->opt.v[OPT_SIDE_CONNECT] : (h->ctx.size() <= 2 ? 1 : 0);
    h->side_only_alone = h->opt.v[OPT_SIDE_CONNECT] < 0 && h->ctx.size() > 1;
    for (FrameCtx &c : h->ctx) {
        if (h->side_connect && !c.side) HIP_TRY(h, hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
        if (!h->side_connect && c.side) {
            (void)hipStreamSynchronize(c.side);
            (void)hipStreamDestroy(c.side);
            c.side = nullptr;
        }
        c.gstack_side = nullptr; // (frame-sized: freed above, made again below when there is a side stream)
    }
    const size_t cap = (size_t)h->npix_padded * mb;
    h->path_capacity = cap;
    int rc;
    for (FrameCtx &c : h->ctx) {
        if ((rc = dev_alloc(h, &c.ps.ray_o, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.ps.ray_d, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.ps.thr, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.ps.illum, cap, nullptr))) return rc;
        c.ps.alpha_rng = nullptr;
        if (h->rng_variant != RPTR_RNG_VARIANT_UNIFORM && (rc = dev_alloc(h, &c.ps.alpha_rng, cap, nullptr))) return rc;
        c.ps.footprint = nullptr; // scenes with textures: set_scene allocates it; a scene set before this call keeps its flags
        if ((h->uses_textures || h->uses_alpha) && (rc = dev_alloc(h, &c.ps.footprint, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.ps.hit_tuv, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.ps.hit_ids, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.sq.d, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.sq.contrib, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.sq.ids, cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.queue[0], cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.queue[1], cap, nullptr))) return rc;
        if ((rc = dev_alloc(h, &c.counters, 1, nullptr))) return rc;
    }
    const size_t npix_local = (size_t)h->width * std::max(h->local_rows, 1);
    if ((rc = dev_alloc(h, &h->accum, npix_local, nullptr))) return rc;
    if ((rc = dev_alloc(h, &h->fb, npix_local, nullptr))) return rc;
    HIP_TRY(h, hipMemsetAsync(h->accum, 0, npix_local * sizeof(float4), h->stream));
    HIP_TRY(h, hipMemsetAsync(h->fb, 0, npix_local * sizeof(uchar4), h->stream));
    if (h->ctx.size() > 1)
        for (FrameCtx &c : h->ctx) {
            const size_t nout = npix_local * (size_t)h->max_batch_frames;
            if ((rc = dev_alloc(h, &c.out_accum, nout, nullptr))) return rc;
            if ((rc = dev_alloc(h, &c.out_fb, nout, nullptr))) return rc;
            HIP_TRY(h, hipMemsetAsync(c.out_accum, 0, nout * sizeof(float4), h->stream));
            HIP_TRY(h, hipMemsetAsync(c.out_fb, 0, nout * sizeof(uchar4), h->stream));
        }
    h->output_ctx = -1;
    h->last_resolved = nullptr;
    h->aov_ctx = 0;
    h->output_overwritten = h->aov_overwritten = false;
    if (h->aovs)
        for (FrameCtx &c : h->ctx)
            for (int k = 0; k < 3; ++k) {
                if ((rc = dev_alloc(h, &c.aov[k], npix_local, nullptr))) return rc;
                HIP_TRY(h, hipMemsetAsync(c.aov[k], 0, npix_local * sizeof(uint2), h->stream));
            }
    // persistent traversal kernels: as many blocks as are co-resident
    int occ = 0;
    HIP_TRY(h, rp_extend_blocks_per_cu(&occ));
    occ = std::max(1, std::min(occ, 8));
    // (handles with TWO frame contexts -- the reference's swap buffers -- keep the full-size grids as well: a frame submitted while no other
    // is in flight, the synchronous loop, is launched like a frame of a one-context handle; side_only_alone above)
    const bool keep_alone = h->ctx.size() == 2 && h->opt.v[OPT_BLOCKS_PER_CU] <= 0;
    h->alone_blocks[0] = keep_alone ? h->num_cus * occ : 0;
    // frames in flight share the CUs: with n contexts a traversal launch asks for about 12 / n blocks per CU instead of all that
    // fit, so that the kernels of the other frames find room next to it (measured, profiles/r01_notes.md: 3 contexts 5 -> 4 blocks
    // 1.50 -> 1.49 ms per full frame; 11 contexts 5 -> 1 blocks 0.30 -> 0.25 ms per 1/8 frame)
    if (h->ctx.size() > 1) occ = std::max(1, std::min(occ, (int)((12 + h->ctx.size() / 2) / h->ctx.size())));
    if (h->opt.v[OPT_BLOCKS_PER_CU] > 0) occ = (int)h->opt.v[OPT_BLOCKS_PER_CU];
    h->persistent_blocks = h->num_cus * occ;
    // the shadow-ray kernels may be compiled for more waves per SIMD than the closest-hit kernels (RP_CONNECT_WAVES): their launches get the
    // blocks THEY can have resident (the same cap with frames in flight)
    for (int sg = 0; sg < 2; ++sg) { // [0]: two-level scenes, [1]: scenes with one instance record (the instantiation compiled for six waves)
        int occ_c = 0;
        HIP_TRY(h, rp_connect_blocks_per_cu(sg, &occ_c));
        occ_c = std::max(1, std::min(occ_c, 8));
        h->alone_blocks[2 + sg] = keep_alone ? h->num_cus * occ_c : 0;
        if (h->ctx.size() > 1) occ_c = std::max(1, std::min(occ_c, (int)((12 + h->ctx.size() / 2) / h->ctx.size())));
        if (h->opt.v[OPT_BLOCKS_PER_CU] > 0) occ_c = (int)h->opt.v[OPT_BLOCKS_PER_CU];
        h->connect_blocks[sg] = h->num_cus * occ_c;
    }
    int occ_l = 0;
    HIP_TRY(h, rp_extend_later_blocks_per_cu(&occ_l));
    occ_l = std::max(1, std::min(occ_l, 8));
    h->alone_blocks[1] = keep_alone ? h->num_cus * occ_l : 0;
    if (h->ctx.size() > 1) occ_l = std::max(1, std::min(occ_l, (int)((12 + h->ctx.size() / 2) / h->ctx.size())));
    if (h->opt.v[OPT_BLOCKS_PER_CU] > 0) occ_l = (int)h->opt.v[OPT_BLOCKS_PER_CU];
    h->extend_later_blocks = h->num_cus * occ_l;
    h->tail_blocks = h->num_cus; // one block per CU (the tail kernel's LDS: two traversal stacks + the shade buffers)
    const size_t stack_threads = (size_t)std::max(std::max(std::max(h->persistent_blocks, h->extend_later_blocks), std::max(h->connect_blocks[0], h->connect_blocks[1])),
                                                  std::max(std::max(h->alone_blocks[0], h->alone_blocks[1]), std::max(h->alone_blocks[2], h->alone_blocks[3]))) * RP_TRAVERSE_BLOCK;
    for (FrameCtx &c : h->ctx) {
        c.gstack_threads = stack_threads;
        if ((rc = dev_alloc(h, &c.gstack, stack_threads * RPTR_BVH_STACK_DEPTH, nullptr))) return rc;
        if (c.side && (rc = dev_alloc(h, &c.gstack_side, stack_threads * RPTR_BVH_STACK_DEPTH, nullptr))) return rc;
    }
    h->frame_id = 0;
    h->frame_offset = 0;
    h->accumulated_spp = 0;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return RPTR_OK;
}

int rptr_hip_set_params(rptr_hip_t *h, const RptrRenderParams *params, const RptrSceneParams *scene_params,
                        const RptrLightSamplingConfig *lighting_params) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (params) {
        if (params->max_path_depth < 1 || params->max_path_depth > RP_MAX_BOUNCES) return fail(h, RPTR_E_INVALID, "max_path_depth out of range");
        h->params = *params;
    }
    if (scene_params) h->scene_params = *scene_params;
    if (lighting_params) {
        if (lighting_params->bin_size < 1 || lighting_params->bin_size > RPTR_BINNED_LIGHTS_BIN_MAX_SIZE)
            return fail(h, RPTR_E_INVALID, "bin_size must be in [1,%d]", RPTR_BINNED_LIGHTS_BIN_MAX_SIZE);
        h->lighting = *lighting_params;
    }
    h->have_params = true;
    return RPTR_OK;
}

#include "host_scene.inl"
#include "host_frame.inl"
#include "host_access.inl"

} // extern "C"

#include "host_comm.h"
