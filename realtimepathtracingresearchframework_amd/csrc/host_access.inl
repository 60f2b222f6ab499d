// host_access.inl -- everything else a host reaches through the handle: BVH policy, freeze frame, point sets, stage timing, the option calls, frame buffer / tile /
// AOV read-backs, ray queries (rptr_hip_trace*), the exported tree
// Part of the ONE translation unit rptr_hip.hip (included there, in this order: host_state.h, host_bvh.inl, host_scene.inl,
// host_frame.inl, host_access.inl, host_comm.h): the host runtime split along its seams; no symbol changed.
int rptr_hip_set_bvh_policy(rptr_hip_t *h, int force_bvh_rebuild, int rebuild_triangle_budget) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (rebuild_triangle_budget < 0) return fail(h, RPTR_E_INVALID, "rebuild_triangle_budget must be >= 0");
    h->bvh_force_rebuild = force_bvh_rebuild != 0;
    h->bvh_budget = rebuild_triangle_budget;
    return RPTR_OK;
}

int rptr_hip_bvh_rebuild_count(const rptr_hip_t *h, uint64_t *out_rebuilds) {
    if (!h || !out_rebuilds) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    *out_rebuilds = h->rebuilds_done;
    return RPTR_OK;
}

int rptr_hip_set_freeze_frame(rptr_hip_t *h, int freeze_frame) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    h->freeze_frame = freeze_frame != 0;
    return RPTR_OK;
}

int rptr_hip_set_rng_variant(rptr_hip_t *h, int rng_variant, const void *table, size_t table_bytes) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (rng_variant < RPTR_RNG_VARIANT_UNIFORM || rng_variant > RPTR_RNG_VARIANT_Z_SBL)
        return fail(h, RPTR_E_INVALID, "rng_variant %d (0 uniform, 1 blue noise, 2 Sobol, 3 Z-Sobol)", rng_variant);
    size_t need = 0;
    if (rng_variant == RPTR_RNG_VARIANT_BN) need = RPTR_BN_TABLE_MIN_BYTES;
    if (rng_variant == RPTR_RNG_VARIANT_SOBOL || rng_variant == RPTR_RNG_VARIANT_Z_SBL) need = RPTR_SOBOL_TABLE_BYTES;
    if (need && (!table || table_bytes < need))
        return fail(h, RPTR_E_INVALID, "rng_variant %d needs a table of %zu bytes (got %zu)", rng_variant, need, table ? table_bytes : (size_t)0);
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = drain(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->rng_table) {
        (void)hipFree(h->rng_table);
        h->rng_table = nullptr;
    }
    if (need) {
        HIP_TRY(h, hipMalloc((void **)&h->rng_table, need));
        HIP_TRY(h, hipMemcpy(h->rng_table, table, need, hipMemcpyHostToDevice));
        for (FrameCtx &c : h->ctx) // the alpha-test generator of closest-hit queries gets its own slot in the path state
            if (!c.ps.alpha_rng && h->path_capacity && (rc = dev_alloc(h, &c.ps.alpha_rng, h->path_capacity, nullptr))) return rc;
    }
    h->rng_variant = rng_variant;
    return RPTR_OK;
}

int rptr_hip_set_stage_timing(rptr_hip_t *h, int level) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (level < 0 || level > 2) return fail(h, RPTR_E_INVALID, "stage timing level %d (0 none, 1 extend only, 2 all stages)", level);
    h->opt.v[OPT_STAGE_TIMING] = level;
    h->stage_timing = level;
    return RPTR_OK;
}

// ---- options (the table at the top of this file)
int rptr_hip_set_option(rptr_hip_t *h, const char *key, int64_t value) {
    const int k = find_option(key);
    if (k < 0) return fail(h, RPTR_E_INVALID, "rptr_hip_set_option: unknown option \"%s\"", key ? key : "(null)");
    if (value < g_opt_desc[k].lo || value > g_opt_desc[k].hi)
        return fail(h, RPTR_E_INVALID, "rptr_hip_set_option: %s = %lld is outside [%lld, %lld]", key, (long long)value, g_opt_desc[k].lo, g_opt_desc[k].hi);
    if (!h) { // the process default: what new handles (and the handle-less rptr_hip_build_bvh_host) start from
        set_process_default_option(k, value);
        return RPTR_OK;
    }
    if (h->opt.from_env[k]) return RPTR_OK; // the environment variable of this option is set: the experimenter's override stands (rptr_hip_get_option tells)
    h->opt.v[k] = value;
    sync_options(h);
    return RPTR_OK;
}
int rptr_hip_get_option(const rptr_hip_t *h, const char *key, int64_t *out_value) {
    if (h && key && out_value && !strcmp(key, "bvh_rebuild_failures")) { // (read-only: a counter, not a switch)
        *out_value = (int64_t)h->rebuild_failures;
        return RPTR_OK;
    }
    if (h && key && out_value && !strcmp(key, "sample_slots")) { // (read-only: the sample slots a frame context holds once initialize has sized
        *out_value = (int64_t)h->max_batch_spp;                  // the path state -- "max_batch_spp" or what the budget allows; 0 before initialize)
        return RPTR_OK;
    }
    const int k = find_option(key);
    if (k < 0 || !out_value) return fail(nullptr, RPTR_E_INVALID, "rptr_hip_get_option: unknown option \"%s\" or NULL result", key ? key : "(null)");
    *out_value = h ? h->opt.v[k] : effective_default_options().v[k];
    return RPTR_OK;
}
int rptr_hip_option_count(void) { return OPT_PUBLIC_COUNT; }
const char *rptr_hip_option_name(int index) { return index >= 0 && index < OPT_PUBLIC_COUNT ? g_opt_desc[index].key : nullptr; }

int rptr_hip_get_framebuffer_size(const rptr_hip_t *h, uint32_t out_whc[3]) {
    if (!h || !out_whc) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    out_whc[0] = (uint32_t)h->width;
    out_whc[1] = (uint32_t)h->height;
    out_whc[2] = 4;
    return RPTR_OK;
}

int rptr_hip_tile_rows(const rptr_hip_t *h, int rank, int32_t *first_and_count, int cap) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    const int n_stripes = (h->height + h->stripe_rows - 1) / h->stripe_rows;
    int n = 0;
    for (int s = rank; s < n_stripes; s += h->world) {
        if (first_and_count && n < cap) {
            first_and_count[2 * n] = s * h->stripe_rows;
            first_and_count[2 * n + 1] = std::min(h->stripe_rows, h->height - s * h->stripe_rows);
        }
        ++n;
    }
    return n;
}

int rptr_hip_local_pixel_count(const rptr_hip_t *h, uint64_t *out_pixels) {
    if (!h || !out_pixels) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    *out_pixels = (uint64_t)h->width * (uint64_t)h->local_rows;
    return RPTR_OK;
}

int rptr_hip_copy_tile_to_device(rptr_hip_t *h, void *device_dst, size_t n_bytes) {
    if (!h || !device_dst) return fail(h, RPTR_E_INVALID, "NULL argument");
    const size_t need = (size_t)h->width * h->local_rows * sizeof(float4);
    if (n_bytes < need) return fail(h, RPTR_E_INVALID, "destination too small: %zu < %zu", n_bytes, need);
    if (h->output_overwritten)
        return fail(h, RPTR_E_INVALID, "the image of the last waited frame is being overwritten by a newer frame in flight on the same frame context: "
                                       "read back before submitting that frame, or rptr_hip_wait for it first");
    HIP_TRY(h, hipSetDevice(h->device));
    // frames in flight: the image of the frame that was waited for last (its context keeps a copy)
    const size_t out_stride = (size_t)h->width * (size_t)std::max(h->local_rows, 1);
    const float4 *src = h->output_ctx >= 0 ? h->ctx[(size_t)h->output_ctx].out_accum + (size_t)h->output_index * out_stride : h->accum;
    if (need) HIP_TRY(h, hipMemcpyAsync(device_dst, src, need, hipMemcpyDeviceToDevice, h->stream));
    return RPTR_OK;
}

extern "C++" {
template <class T>
static int readback_rows(rptr_hip *h, const T *dev_local, T *host_full, size_t n_elems_host) {
    const size_t need = (size_t)h->width * h->height;
    if (n_elems_host < need) return fail(h, RPTR_E_INVALID, "read-back buffer too small");
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<T> tmp((size_t)h->width * std::max(h->local_rows, 1));
    if (h->local_rows)
        HIP_TRY(h, hipMemcpyAsync(tmp.data(), dev_local, (size_t)h->width * h->local_rows * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const int n_stripes = (h->height + h->stripe_rows - 1) / h->stripe_rows;
    int local_row = 0;
    for (int s = h->rank; s < n_stripes; s += h->world) {
        const int first = s * h->stripe_rows, cnt = std::min(h->stripe_rows, h->height - first);
        memcpy(host_full + (size_t)first * h->width, tmp.data() + (size_t)local_row * h->width, (size_t)cnt * h->width * sizeof(T));
        local_row += cnt;
    }
    return RPTR_OK;
}
} // extern "C++"

int rptr_hip_readback_f32(rptr_hip_t *h, float *rgba, size_t n_floats) {
    if (!h || !rgba) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (h->output_overwritten)
        return fail(h, RPTR_E_INVALID, "the image of the last waited frame is being overwritten by a newer frame in flight on the same frame context: "
                                       "read back before submitting that frame, or rptr_hip_wait for it first");
    const size_t out_stride = (size_t)h->width * (size_t)std::max(h->local_rows, 1);
    return readback_rows<float4>(h, h->output_ctx >= 0 ? h->ctx[(size_t)h->output_ctx].out_accum + (size_t)h->output_index * out_stride : h->accum,
                                 reinterpret_cast<float4 *>(rgba), n_floats / 4);
}
int rptr_hip_readback_u8(rptr_hip_t *h, unsigned char *rgba, size_t n_bytes) {
    if (!h || !rgba) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (h->output_overwritten)
        return fail(h, RPTR_E_INVALID, "the image of the last waited frame is being overwritten by a newer frame in flight on the same frame context: "
                                       "read back before submitting that frame, or rptr_hip_wait for it first");
    const size_t out_stride = (size_t)h->width * (size_t)std::max(h->local_rows, 1);
    const uchar4 *src = h->output_ctx >= 0 ? h->ctx[(size_t)h->output_ctx].out_fb + (size_t)h->output_index * out_stride : h->fb;
    if (h->params.render_upscale_factor != 2) return readback_rows<uchar4>(h, src, reinterpret_cast<uchar4 *>(rgba), n_bytes / 4);
    // render_upscale_factor == 2 (process_samples.comp:192-197): the frame buffer has twice the render resolution, every rendered
    // pixel fills a 2x2 block. Replicated here, on the way out (rows of other ranks stay untouched, as in the 1:1 read-back).
    const size_t W = (size_t)h->width, H = (size_t)h->height;
    if (n_bytes / 4 < 4 * W * H) return fail(h, RPTR_E_INVALID, "read-back buffer too small for the 2x upscaled frame buffer");
    std::vector<uchar4> lo(W * H);
    const uchar4 *big = reinterpret_cast<const uchar4 *>(rgba);
    for (size_t y = 0; y < H; ++y) // keep what the caller's buffer holds for rows this rank does not own
        for (size_t x = 0; x < W; ++x) lo[y * W + x] = big[(2 * y) * (2 * W) + 2 * x];
    int rc = readback_rows<uchar4>(h, src, lo.data(), lo.size());
    if (rc) return rc;
    uchar4 *out = reinterpret_cast<uchar4 *>(rgba);
    for (size_t y = 0; y < H; ++y)
        for (size_t x = 0; x < W; ++x) {
            const uchar4 px = lo[y * W + x];
            out[(2 * y) * (2 * W) + 2 * x] = out[(2 * y) * (2 * W) + 2 * x + 1] = out[(2 * y + 1) * (2 * W) + 2 * x] = out[(2 * y + 1) * (2 * W) + 2 * x + 1] = px;
        }
    return RPTR_OK;
}

int rptr_hip_readback_aov(rptr_hip_t *h, int aov_index, uint16_t *rgba16f, size_t n_halfs) {
    if (!h || !rgba16f) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (aov_index < 0 || aov_index >= 3) return fail(h, RPTR_E_INVALID, "AOV index %d (0 albedo+roughness, 1 normal+depth, 2 motion+jitter)", aov_index);
    if (h->aov_overwritten)
        return fail(h, RPTR_E_INVALID, "the AOV images of the last finished frame are being overwritten by a newer frame in flight on the same frame "
                                       "context: read back before submitting that frame, or rptr_hip_wait for it first");
    const FrameCtx &c = h->ctx[(size_t)h->aov_ctx];
    if (!c.aov[aov_index]) return fail(h, RPTR_E_INVALID, "AOV images are switched off (RPTR_AOVS=0) or initialize() has not run");
    return readback_rows<uint2>(h, c.aov[aov_index], reinterpret_cast<uint2 *>(rgba16f), n_halfs / 4);
}

int rptr_hip_trace(rptr_hip_t *h, const RptrRenderRayQuery *queries, int n, float *out4) {
    return rptr_hip_trace_counted(h, queries, n, out4, nullptr, nullptr, 0);
}

int rptr_hip_trace_counted(rptr_hip_t *h, const RptrRenderRayQuery *queries, int n, float *out4, uint32_t *visits2, const float *tmin, int any_hit) {
    if (!h || !queries || !out4 || n < 0) return fail(h, RPTR_E_INVALID, "bad argument");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "trace before set_scene");
    if (!h->ctx[0].gstack) return fail(h, RPTR_E_INVALID, "trace before initialize");
    {
        int rc0 = drain(h); // the query kernel borrows context 0's cursor and stack scratch
        if (rc0) return rc0;
        if ((rc0 = ensure_master_tree(h))) return rc0;
    }
    if (n == 0) return RPTR_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    RptrRenderRayQuery *dq = nullptr;
    float4 *dr = nullptr;
    uint2 *dv = nullptr;
    float *dt = nullptr;
    HIP_TRY(h, hipMalloc((void **)&dq, (size_t)n * sizeof(RptrRenderRayQuery)));
    if (hipMalloc((void **)&dr, (size_t)n * sizeof(float4)) != hipSuccess || (visits2 && hipMalloc((void **)&dv, (size_t)n * sizeof(uint2)) != hipSuccess) ||
        (tmin && hipMalloc((void **)&dt, (size_t)n * sizeof(float)) != hipSuccess)) {
        (void)hipFree(dq);
        (void)hipFree(dr);
        (void)hipFree(dv);
        return fail(h, RPTR_E_NOMEM, "hipMalloc failed");
    }
    int rc = RPTR_OK;
    do {
        if (hipMemcpyAsync(dq, queries, (size_t)n * sizeof(RptrRenderRayQuery), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(dr, out4, (size_t)n * sizeof(float4), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            (tmin && hipMemcpyAsync(dt, tmin, (size_t)n * sizeof(float), hipMemcpyHostToDevice, h->stream) != hipSuccess)) {
            rc = fail(h, RPTR_E_HIP, "upload failed");
            break;
        }
        // cursor_extend doubles as the pool cursor of the query kernel (same stream, no overlap with a frame)
        hipLaunchKernelGGL(rp_k_reset_u32, dim3(1), dim3(1), 0, h->stream, &h->ctx[0].counters->bounce[0].cursor_extend);
        auto launch = [&](auto kernel) {
            hipLaunchKernelGGL(kernel, dim3(h->persistent_blocks), dim3(RP_TRAVERSE_BLOCK), 0, h->stream, h->master.dscene, dq, (uint32_t)n, dr,
                               &h->ctx[0].counters->bounce[0].cursor_extend, h->ctx[0].gstack, dv, dt);
        };
        pick(h->master.dscene.single_instance != 0, [&](auto S) {
            if (any_hit)
                launch(rp_k_trace<true, true, decltype(S)::value>);
            else if (visits2)
                launch(rp_k_trace<true, false, decltype(S)::value>);
            else
                launch(rp_k_trace<false, false, decltype(S)::value>);
        });
        if ((visits2 && hipMemcpyAsync(visits2, dv, (size_t)n * sizeof(uint2), hipMemcpyDeviceToHost, h->stream) != hipSuccess) ||
            hipMemcpyAsync(out4, dr, (size_t)n * sizeof(float4), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess || hipGetLastError() != hipSuccess) {
            rc = fail(h, RPTR_E_HIP, "trace kernel failed");
            break;
        }
    } while (0);
    (void)hipFree(dq);
    (void)hipFree(dr);
    (void)hipFree(dv);
    (void)hipFree(dt);
    return rc;
}

extern "C++" {
// the RQ_CLOSEST kernel over DEVICE buffers, asynchronously on `st`
static int trace_device_on(rptr_hip *h, const RptrRenderRayQuery *dq, int n, float4 *dr, hipStream_t st) {
    if (n == 0) return RPTR_OK;
    hipLaunchKernelGGL(rp_k_reset_u32, dim3(1), dim3(1), 0, st, &h->ctx[0].counters->bounce[0].cursor_extend);
    pick(h->master.dscene.single_instance != 0, [&](auto S) {
        hipLaunchKernelGGL((rp_k_trace<false, false, decltype(S)::value>), dim3(h->persistent_blocks), dim3(RP_TRAVERSE_BLOCK), 0, st, h->master.dscene, dq, (uint32_t)n, dr,
                           &h->ctx[0].counters->bounce[0].cursor_extend, h->ctx[0].gstack, (uint2 *)nullptr, (const float *)nullptr);
    });
    HIP_TRY(h, hipGetLastError());
    return RPTR_OK;
}
}

int rptr_hip_trace_device(rptr_hip_t *h, const RptrRenderRayQuery *device_queries, int n, float *device_out4, void *hip_stream) {
    if (!h || !device_queries || !device_out4 || n < 0) return fail(h, RPTR_E_INVALID, "bad argument");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "trace before set_scene");
    if (!h->ctx[0].gstack) return fail(h, RPTR_E_INVALID, "trace before initialize");
    int rc = drain(h); // the query kernel borrows context 0's cursor and stack scratch
    if (rc || (rc = ensure_master_tree(h))) return rc;
    HIP_TRY(h, hipSetDevice(h->device));
    hipStream_t st = hip_stream ? (hipStream_t)hip_stream : h->stream;
    if (st != h->stream) { // the caller's stream sees the scene uploads / refits queued on the backend's, and later frames see the queries
        hipEvent_t e;
        HIP_TRY(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        (void)hipEventRecord(e, h->stream);
        (void)hipStreamWaitEvent(st, e, 0);
        rc = trace_device_on(h, device_queries, n, reinterpret_cast<float4 *>(device_out4), st);
        (void)hipEventRecord(e, st);
        (void)hipStreamWaitEvent(h->stream, e, 0);
        (void)hipEventDestroy(e);
        return rc;
    }
    return trace_device_on(h, device_queries, n, reinterpret_cast<float4 *>(device_out4), st);
}

int rptr_hip_enable_ray_queries(rptr_hip_t *h, int max_queries, int max_queries_per_pixel, void **out_device_queries, void **out_device_results) {
    if (!h || max_queries < 0 || max_queries_per_pixel < 0) return fail(h, RPTR_E_INVALID, "bad argument");
    HIP_TRY(h, hipSetDevice(h->device));
    // vulkan/render_vulkan.cpp:430-455: max(fixed budget, per-pixel budget x frame size) queries of 32 bytes, as many float4 results
    const size_t want = std::max<size_t>((size_t)max_queries, (size_t)h->width * (size_t)h->height * (size_t)max_queries_per_pixel);
    if (want > h->rq_capacity) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->rq_queries) (void)hipFree(h->rq_queries);
        if (h->rq_results) (void)hipFree(h->rq_results);
        h->rq_queries = nullptr;
        h->rq_results = nullptr;
        h->rq_capacity = 0;
        if (hipMalloc((void **)&h->rq_queries, want * sizeof(RptrRenderRayQuery)) != hipSuccess || hipMalloc((void **)&h->rq_results, want * sizeof(float4)) != hipSuccess) {
            if (h->rq_queries) (void)hipFree(h->rq_queries);
            h->rq_queries = nullptr;
            return fail(h, RPTR_E_NOMEM, "hipMalloc of the ray query buffers (%zu queries) failed", want);
        }
        h->rq_capacity = want;
    }
    if (out_device_queries) *out_device_queries = h->rq_queries;
    if (out_device_results) *out_device_results = h->rq_results;
    return RPTR_OK;
}

int rptr_hip_render_ray_queries(rptr_hip_t *h, int num_queries) {
    if (!h || num_queries < 0) return fail(h, RPTR_E_INVALID, "bad argument");
    if ((size_t)num_queries > h->rq_capacity) return fail(h, RPTR_E_INVALID, "%d ray queries exceed the budget of %zu (rptr_hip_enable_ray_queries)", num_queries, h->rq_capacity);
    return rptr_hip_trace_device(h, h->rq_queries, num_queries, reinterpret_cast<float *>(h->rq_results), nullptr);
}

int rptr_hip_set_light_sampling_variant(rptr_hip_t *h, int variant) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (variant != 0 && variant != 1) return fail(h, RPTR_E_INVALID, "unknown light sampling variant %d (0 = NONE, 1 = RIS)", variant);
    h->lights_disabled = variant == 0;
    return RPTR_OK;
}

int rptr_hip_build_bvh_host(const RptrSceneDesc *scene, void *nodes, size_t *n_nodes, void *tris, size_t *n_tris, void *instances,
                            size_t *n_instances, int32_t *out_stack_need) {
    if (!scene) return fail(nullptr, RPTR_E_INVALID, "NULL scene");
    {
        const std::string bad = validate_scene_tables(scene);
        if (!bad.empty()) return fail(nullptr, RPTR_E_INVALID, "%s", bad.c_str());
    }
    HostBvh B;
    build_host_bvh(scene, B, effective_default_options());
    if (nodes && n_nodes && *n_nodes >= B.nodes.size()) memcpy(nodes, B.nodes.data(), B.nodes.size() * sizeof(RptrBvh4Node));
    if (tris && n_tris && *n_tris >= B.tris.size()) memcpy(tris, B.tris.data(), B.tris.size() * sizeof(RptrBvhTri));
    if (instances && n_instances && *n_instances >= B.insts.size()) memcpy(instances, B.insts.data(), B.insts.size() * sizeof(RptrBvhInstance));
    if (n_nodes) *n_nodes = B.nodes.size();
    if (n_tris) *n_tris = B.tris.size();
    if (n_instances) *n_instances = B.insts.size();
    if (out_stack_need) *out_stack_need = B.stack_need;
    return RPTR_OK;
}

int rptr_hip_export_bvh(rptr_hip_t *h, void *nodes, size_t *n_nodes, void *tris, size_t *n_tris, void *instances, size_t *n_instances) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "export before set_scene");
    {
        int rc0 = ensure_master_tree(h);
        if (rc0) return rc0;
    }
    if (h->host_bvh_stale) { // a refit happened on the device: refresh the host mirror first
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, hipMemcpy(h->h_nodes.data(), h->master.dscene.nodes, h->h_nodes.size() * sizeof(RptrBvh4Node), hipMemcpyDeviceToHost));
        if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(h->h_tris.data(), h->master.dscene.tris, h->h_tris.size() * sizeof(RptrBvhTri), hipMemcpyDeviceToHost));
        h->host_bvh_stale = false;
    }
    if (nodes && n_nodes && *n_nodes >= h->h_nodes.size()) memcpy(nodes, h->h_nodes.data(), h->h_nodes.size() * sizeof(RptrBvh4Node));
    if (tris && n_tris && *n_tris >= h->h_tris.size()) memcpy(tris, h->h_tris.data(), h->h_tris.size() * sizeof(RptrBvhTri));
    if (instances && n_instances && *n_instances >= h->h_insts.size())
        memcpy(instances, h->h_insts.data(), h->h_insts.size() * sizeof(RptrBvhInstance));
    if (n_nodes) *n_nodes = h->h_nodes.size();
    if (n_tris) *n_tris = h->h_tris.size();
    if (n_instances) *n_instances = h->h_insts.size();
    return RPTR_OK;
}

