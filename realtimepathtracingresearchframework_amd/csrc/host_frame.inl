// host_frame.inl -- a frame: camera basis, the stage launches of a launch sequence (render_batch_impl), collecting a frame (finish_frame), wait / render / stats
// Part of the ONE translation unit rptr_hip.hip (included there, in this order: host_state.h, host_bvh.inl, host_scene.inl,
// host_frame.inl, host_access.inl, host_comm.h): the host runtime split along its seams; no symbol changed.
// host part of a3: vulkan/render_vulkan.cpp:2880-2896
static void compute_view(const RptrCamera &c, int W, int H, RpFrame &f) {
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    auto normalize = [](float v[3]) {
        float inv = 1.0f / sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        v[0] *= inv;
        v[1] *= inv;
        v[2] *= inv;
    };
    const float plane_y = 2.f * tanf((0.5f * c.fovy) * 0.01745329251994329576923690768489f);
    const float aspect = static_cast<float>(W) / H;
    const float plane_x = plane_y * aspect;
    float du[3], dv[3];
    cross(c.dir, c.up, du);
    normalize(du);
    for (int k = 0; k < 3; ++k) du[k] *= plane_x;
    cross(du, c.dir, dv);
    normalize(dv);
    for (int k = 0; k < 3; ++k) dv[k] = -dv[k] * plane_y;
    for (int k = 0; k < 3; ++k) {
        f.cam_pos[k] = c.pos[k];
        f.cam_du[k] = du[k];
        f.cam_dv[k] = dv[k];
        f.cam_dir_top_left[k] = c.dir[k] - 0.5f * du[k] - 0.5f * dv[k];
    }
}

// x / y / w rows of VP (render_vulkan.cpp:2926-2931): inverse of the camera-to-world matrix with columns cross(dir, up), up, -dir,
// pos; glm::infinitePerspective(radians(fovy), aspect, 0.5f) contributes P00 and P11 (GLM's published formulas)
static void compute_view_projection(const RptrCamera &c, int W, int H, float view[12], float proj[2]) {
    auto cross = [](const float a[3], const float b[3], float o[3]) {
        o[0] = a[1] * b[2] - b[1] * a[2];
        o[1] = a[2] * b[0] - b[2] * a[0];
        o[2] = a[0] * b[1] - b[0] * a[1];
    };
    auto dot = [](const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; };
    float cx[3], cz[3] = {-c.dir[0], -c.dir[1], -c.dir[2]}, r[3][3];
    cross(c.dir, c.up, cx);
    cross(c.up, cz, r[0]);
    cross(cz, cx, r[1]);
    cross(cx, c.up, r[2]);
    const float inv_det = 1.0f / dot(cx, r[0]);
    for (int k = 0; k < 3; ++k) {
        for (int j = 0; j < 3; ++j) r[k][j] *= inv_det;
        view[4 * k + 0] = r[k][0];
        view[4 * k + 1] = r[k][1];
        view[4 * k + 2] = r[k][2];
        view[4 * k + 3] = -dot(r[k], c.pos);
    }
    const float z_near = 0.5f, aspect = static_cast<float>(W) / H;
    const float range = tanf((c.fovy * 0.01745329251994329576923690768489f) / 2.0f) * z_near;
    const float left = -range * aspect, right = range * aspect, bottom = -range, top = range;
    proj[0] = (2.0f * z_near) / (right - left);
    proj[1] = (2.0f * z_near) / (top - bottom);
}

extern "C++" {
// runtime flag -> template argument: f(std::true_type) or f(std::false_type)
template <class F>
static inline void pick(bool v, F &&f) {
    if (v)
        f(std::true_type());
    else
        f(std::false_type());
}

static void launch_shade(rptr_hip *h, FrameCtx &c, int variant, const RpScene &scene, const RpFrame &f, const uint32_t *order, int bounce, int out) {
    // without emissive triangles and with all NEE probability on the sun the light-sampling branch is dead code
    const bool lights = (h->num_lights > 0 && !h->lights_disabled) || f.sp.sun_radiance[3] < 1.0f;
    const RpLaunch l = {(unsigned)grid_for(h, h->path_capacity), c.stream, nullptr, nullptr};
    rp_launch_shade(variant, h->opt.v[OPT_FAST_MATH] != 0, l, bounce == 0, lights, h->uses_textures,
                    f.rng_variant != RPTR_RNG_VARIANT_UNIFORM || f.rp.enable_raster_taa != 0, scene, f, c.ps, c.sq, order,
                    (const uint32_t *)&c.counters->bounce[bounce].queue_count, c.queue[out], &c.counters->bounce[bounce + 1].queue_count,
                    &c.counters->bounce[bounce].shadow_count, c.counters);
}

static void add_counters(RpCounters &dst, const RpCounters &c) {
    dst.rays_closest += c.rays_closest;
    dst.rays_shadow += c.rays_shadow;
    dst.nodes += c.nodes;
    dst.tris += c.tris;
    dst.nodes_shadow += c.nodes_shadow;
    dst.tris_shadow += c.tris_shadow;
    dst.hits_shaded += c.hits_shaded;
}

// waits for the frame in flight on `c` and turns its events / counters into RptrStats
// `which`: the frame of the batch that is being collected (-1: all of them, stats dropped)
static int finish_frame(rptr_hip *h, FrameCtx &c, RptrStats *out_stats, int which = -1) {
    if (!c.pending) return fail(h, RPTR_E_INVALID, "no frame in flight on this context");
    const uint32_t all = c.batch_n >= 32 ? ~0u : ((1u << c.batch_n) - 1u);
    if (c.synced) { // a later frame of a batch whose end has been awaited already
        RptrStats st = c.batch_stats;
        st.spp = c.batch_spp_after[std::max(which, 0)];
        h->stats = st;
        if (out_stats) *out_stats = st;
        c.collected |= which < 0 ? all : (1u << which);
        if (c.collected == all) c.pending = false;
        if (h->ctx.size() > 1) {
            h->output_ctx = (int)(&c - h->ctx.data());
            h->output_index = std::max(which, 0);
            h->output_overwritten = false;
        }
        return RPTR_OK;
    }
    // work queued on the backend's stream from here on (tile copies, read-backs) sees this frame; joining at collection
    // time, not at submission, is what lets the next frame's dependency event pass while this frame still runs
    if (h->ctx.size() > 1) HIP_TRY(h, hipStreamWaitEvent(h->stream, c.ev_end, 0));
    HIP_TRY(h, hipEventSynchronize(c.ev_end));
    c.synced = true;
    c.collected |= which < 0 ? all : (1u << which);
    if (c.collected == all) c.pending = false;
#ifdef RP_PROF
    {
        unsigned long long pr[16];
        HIP_TRY(h, rp_prof_exchange(pr)); // (the counters of the traversal kernels live in k_extend.hip's copy of rp_prof)
        fprintf(stderr, "[RP_PROF] node-phase cycles %llu wave-iters %llu lane-iters %llu phases %llu leaf-cycles %llu | cyc/wave-iter %.1f util %.3f iters/phase %.2f leafcyc/phase %.1f\n",
                pr[0], pr[1], pr[2], pr[3], pr[4], double(pr[0]) / double(pr[1] ? pr[1] : 1), double(pr[2]) / (64.0 * double(pr[1] ? pr[1] : 1)),
                double(pr[1]) / double(pr[3] ? pr[3] : 1), double(pr[4]) / double(pr[3] ? pr[3] : 1));
        fprintf(stderr, "[RP_PROF] lost lane-iterations: idle-at-entry %.3f leaf-at-entry %.3f dropped-out %.3f (fractions of 64*wave-iters)\n",
                double(pr[5]) / (64.0 * double(pr[1] ? pr[1] : 1)), double(pr[6]) / (64.0 * double(pr[1] ? pr[1] : 1)),
                double(pr[7]) / (64.0 * double(pr[1] ? pr[1] : 1)));
        fprintf(stderr, "[RP_PROF] time: node %.3g leaf+done %.3g refill %.3g | per phase: tri lanes %.2f (in %.2f of phases) instance lanes %.2f (in %.2f of phases)\n",
                double(pr[0]), double(pr[4]), double(pr[8]), double(pr[9]) / double(pr[3] ? pr[3] : 1), double(pr[11]) / double(pr[3] ? pr[3] : 1),
                double(pr[10]) / double(pr[3] ? pr[3] : 1), double(pr[12]) / double(pr[3] ? pr[3] : 1));
        fprintf(stderr, "[RP_PROF] leaf items: %llu triangle leaves, %llu instance entries (lane counts; per ray: divide by the frame's ray count)\n", pr[9], pr[10]);
        fprintf(stderr, "[RP_PROF] node iterations on the generic stack path (some lane within 3 entries of the end of its LDS stack): %.4f\n",
                double(pr[13]) / double(pr[1] ? pr[1] : 1));
    }
#endif
    RptrStats &st = h->stats;
    memset(&st, 0, sizeof(st));
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c.ev_begin, c.ev_end);
    st.render_time_ms = ms;
    for (const Span &sp : c.spans) {
        float t = 0.f;
        (void)hipEventElapsedTime(&t, sp.a, sp.b);
        if (sp.kind == 0) st.extend_time_ms += t;
        else if (sp.kind == 1) st.connect_time_ms += t;
        else {
            st.shade_time_ms += t;
            if (sp.kind == 2) st.shade_only_time_ms += t;
            else if (sp.kind == 3) st.tail_time_ms += t;
            else if (sp.kind == 4) st.resolve_time_ms += t;
        }
    }
    RpCounters tot = c.earlier_batches;
    if (h->local_rows > 0) add_counters(tot, *c.host_counters);
    st.rays_closest = tot.rays_closest;
    st.rays_shadow = tot.rays_shadow;
    st.nodes_visited = tot.nodes + tot.nodes_shadow;
    st.tris_tested = tot.tris + tot.tris_shadow;
    st.nodes_closest = tot.nodes;
    st.tris_closest = tot.tris;
    st.hits_shaded = tot.hits_shaded;
    st.launches_extend = c.launches_extend;
    st.launches_connect = c.launches_connect;
    st.device_bytes_allocated = h->bytes_allocated;
    if (c.batch_n > 1) { // the frames of a batch share its launches: each reports an equal share
        const float inv = 1.0f / float(c.batch_n);
        st.render_time_ms *= inv;
        st.extend_time_ms *= inv;
        st.connect_time_ms *= inv;
        st.shade_time_ms *= inv;
        st.shade_only_time_ms *= inv;
        st.tail_time_ms *= inv;
        st.resolve_time_ms *= inv;
        for (uint64_t *v : {&st.rays_closest, &st.rays_shadow, &st.nodes_visited, &st.tris_tested, &st.hits_shaded, &st.nodes_closest, &st.tris_closest})
            *v /= (uint64_t)c.batch_n;
    }
    c.batch_stats = st;
    st.spp = c.batch_spp_after[std::max(which, 0)];
    if (h->ctx.size() > 1) {
        h->output_ctx = (int)(&c - h->ctx.data());
        h->output_index = std::max(which, 0);
        h->output_overwritten = false;
    }
    h->aov_ctx = (int)(&c - h->ctx.data());
    h->aov_overwritten = false;
    if (h->local_rows > 0) {
        // where the next frame hands over to the tail kernel: the first bounce whose queue was short in this frame. Queue
        // lengths are known up to the bounce the tail took over at (it does not publish its block-local lists), so the
        // hand-over moves later by one bounce per frame at most
        const int depth = h->params.max_path_depth, used = std::min(c.tail_from, depth);
        int next = depth;
        for (int b = 1; b <= std::min(used, depth - 1); ++b)
            if (c.host_counters->bounce[b].queue_count <= (uint32_t)h->tail_threshold) {
                next = b;
                break;
            }
        if (next == depth && used < depth) // the tail's own queue was long: one bounce later, or (far too long) a frame without a tail to see all queues again
            next = c.host_counters->bounce[used].queue_count > 4u * (uint32_t)h->tail_threshold ? depth : std::min(depth, used + 1);
        h->tail_adaptive = next;
    }
    if (out_stats) *out_stats = st;
    return RPTR_OK;
}

// every frame in flight is waited for (its stats are dropped): before anything that touches shared state
static int drain(rptr_hip *h) {
    for (FrameCtx &c : h->ctx)
        if (c.pending) {
            int rc = finish_frame(h, c, nullptr);
            if (rc) return rc;
        }
    return RPTR_OK;
}
} // extern "C++"

int rptr_hip_render_async(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp, int reset_accumulation, int count_traversal,
                          uint64_t *out_ticket) {
    return rptr_hip_render_batch_async(h, camera, variant, spp, 1, reset_accumulation, 0, count_traversal, out_ticket);
}

extern "C++" {
static int render_batch_impl(rptr_hip_t *h, const RptrCamera *camera, bool per_frame_cameras, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                             int count_traversal, uint64_t *out_tickets);
}
int rptr_hip_render_batch_async(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                                int count_traversal, uint64_t *out_tickets) {
    return render_batch_impl(h, camera, false, variant, spp, n_frames, reset_first, reset_rest, count_traversal, out_tickets);
}
int rptr_hip_render_batch_cameras_async(rptr_hip_t *h, const RptrCamera *cameras, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                                        int count_traversal, uint64_t *out_tickets) {
    return render_batch_impl(h, cameras, n_frames > 1, variant, spp, n_frames, reset_first, reset_rest, count_traversal, out_tickets);
}

extern "C++" {
// camera: ONE camera for all frames of the sequence, or (per_frame_cameras) n_frames of them
// what a render call may ask of this handle (RPTR_E_INVALID with the reason otherwise)
static int check_render_arguments(rptr_hip_t *h, const RptrCamera *camera, bool per_frame_cameras, int variant, int spp, int n_frames) {
    if (!h || !camera) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (n_frames < 1) return fail(h, RPTR_E_INVALID, "n_frames must be >= 1");
    if (per_frame_cameras && n_frames > RP_BATCH_CAMS)
        return fail(h, RPTR_E_INVALID, "a launch sequence holds at most %d frames with cameras of their own", RP_BATCH_CAMS);
    if (n_frames > 1) {
        if (h->ctx.size() < 2) return fail(h, RPTR_E_INVALID, "batches of frames need frames_in_flight >= 2 (every frame of a batch keeps its own image)");
        if (n_frames > h->max_batch_frames) return fail(h, RPTR_E_INVALID, "a batch holds at most %d frames (option \"max_batch_frames\", read by rptr_hip_initialize)", h->max_batch_frames);
        if (n_frames * spp > h->max_batch_spp)
            return fail(h, RPTR_E_INVALID, "%d frames of %d samples do not fit the %d sample slots in flight (RPTR_PATH_BUDGET_MB)", n_frames, spp, h->max_batch_spp);
        if (h->freeze_frame) return fail(h, RPTR_E_INVALID, "a frozen frame cannot be batched with others");
    }
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "render before set_scene");
    if (h->width == 0) return fail(h, RPTR_E_INVALID, "render before initialize");
    if (variant != RPTR_VARIANT_GLTF && variant != RPTR_VARIANT_SIMPLE && variant != RPTR_VARIANT_GLTF_TRANSMISSION)
        return fail(h, RPTR_E_INVALID, "unknown variant %d", variant);
    if (spp < 1) return fail(h, RPTR_E_INVALID, "spp must be >= 1");
    return RPTR_OK;
}
// the frame constants of a launch sequence (RpFrame: render / scene / lighting parameters, the camera basis of frame 0 and -- per_frame_cameras --
// of every frame, the AOV view of the last frame, tiling and divisors, light bins); frame_id / sample bookkeeping is the caller's
static void fill_frame_constants(rptr_hip_t *h, FrameCtx &c, const RptrCamera *camera, bool per_frame_cameras, int variant, int spp, int n_frames, int reset_rest, RpFrame &f) {
    memset(&f, 0, sizeof(f));
    f.rp = h->params;
    f.sp = h->scene_params;
    f.lc = h->lighting;
    compute_view(camera[0], h->width, h->height, f);
    if (per_frame_cameras) { // every frame of the sequence looks through its own camera (kernels.h rp_primary_ray_ex: the general instantiation)
        f.per_frame_cams = 1;
        for (int k = 0; k < n_frames; ++k) {
            RpFrame t;
            compute_view(camera[k], h->width, h->height, t);
            memcpy(f.cams[k].pos, t.cam_pos, sizeof(t.cam_pos));
            memcpy(f.cams[k].du, t.cam_du, sizeof(t.cam_du));
            memcpy(f.cams[k].dv, t.cam_dv, sizeof(t.cam_dv));
            memcpy(f.cams[k].dir_top_left, t.cam_dir_top_left, sizeof(t.cam_dir_top_left));
        }
    }
    {
        // the AOV images are those of the LAST frame of the sequence: its view, and as VP_reference the view of the frame before it (the
        // previous submission's last camera when the sequence is one frame)
        const RptrCamera &last = camera[per_frame_cameras ? n_frames - 1 : 0];
        const RptrCamera &before = n_frames > 1 ? camera[per_frame_cameras ? n_frames - 2 : 0] : (h->have_prev_camera ? h->prev_camera : last);
        compute_view_projection(last, h->width, h->height, f.view, f.proj);
        compute_view_projection(before, h->width, h->height, f.view_ref, f.proj_ref);
        memcpy(f.aov_cam_pos, last.pos, sizeof(f.aov_cam_pos));
        h->prev_camera = last;
        h->have_prev_camera = true;
    }
    f.aov_albedo_roughness = c.aov[0];
    f.aov_normal_depth = c.aov[1];
    f.aov_motion_jitter = c.aov[2];
    f.frame_offset = h->frame_offset;
    f.batch_frames = n_frames;
    f.frame_spp = spp;
    f.batch_reset = reset_rest ? 1 : 0;
    f.div_frame_spp = rp_make_div((uint32_t)spp);
    f.out_stride = (size_t)h->width * (size_t)std::max(h->local_rows, 1);
    f.variant = variant;
    f.width = h->width;
    f.height = h->height;
    f.local_rows = h->local_rows;
    f.tiles_x = h->tiles_x;
    f.tiles_y = h->tiles_y;
    f.npix_padded = h->npix_padded;
    f.rank = h->rank;
    f.world = h->world;
    f.stripe_rows = h->stripe_rows;
    f.div_npix_padded = rp_make_div((uint32_t)h->npix_padded);
    f.div_tiles_x = rp_make_div((uint32_t)(h->tiles_x / RP_TILE_BLOCK));
    f.div_stripe_rows = rp_make_div((uint32_t)h->stripe_rows);
    f.div_width = rp_make_div((uint32_t)h->width);
    f.num_bins = (h->num_lights + (h->lighting.bin_size - 1)) / h->lighting.bin_size;
    if (h->lights_disabled) { // LIGHT_SAMPLING_VARIANT_NONE (rendering/mc/nee.glsl:12-14): every NEE sample goes to the sun
        // the adapter hands sun_radiance.w = 1 with this variant (vulkan/render_sky.cpp:67-70: light_count is 0 without the binned-lights
        // extension); emitters that are HIT keep their full weight: pdf of picking them = (1 - 1) / (bins x solid angle) with a bin count
        // that must not be zero for that product to be 0 rather than NaN
        f.num_bins = std::max(f.num_bins, 1);
        f.sp.sun_radiance[3] = 1.0f;
    }
    // north_star's regrouping of rays by material lives INSIDE the shade kernel's LDS compaction (kernels.h rp_shade_body, RPTR_REGROUP=1):
    // measured on C3 with 48 textured materials it costs 6 % of the shade time and gains nothing (every material runs the same BSDF code),
    // so it is off unless asked for. The separate counting-sort pass of rounds 1-2 (rp_k_sort_*: three launches per bounce, one frame
    // context only, 0.4 ms per frame) lost on every configuration and is gone (profiles/r03_notes.md section 6).
    f.regroup_materials = h->opt.v[OPT_REGROUP] != 0 ? 1 : 0;
}
static int render_batch_impl(rptr_hip_t *h, const RptrCamera *camera, bool per_frame_cameras, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                             int count_traversal, uint64_t *out_tickets) {
    const int reset_accumulation = reset_first;
    {
        const int rc_args = check_render_arguments(h, camera, per_frame_cameras, variant, spp, n_frames);
        if (rc_args) return rc_args;
    }
    HIP_TRY(h, hipSetDevice(h->device));
    FrameCtx &c = h->ctx[(size_t)h->next_ctx];
    if (c.pending)
        return fail(h, RPTR_E_INVALID, "all %zu frames in flight are busy: rptr_hip_wait for ticket %llu first", h->ctx.size(),
                    (unsigned long long)c.ticket);
    h->next_ctx = (h->next_ctx + 1) % (int)h->ctx.size();
    const bool multi = h->ctx.size() > 1;
    if (multi) { // this context's images are about to be rewritten: what was queued on the backend's stream so far still sees the old
                 // ones (ev_dep below), a read-back issued after this submission would not
        if ((int)(&c - h->ctx.data()) == h->output_ctx) h->output_overwritten = true;
        if ((int)(&c - h->ctx.data()) == h->aov_ctx) h->aov_overwritten = true;
    }
    // begin_frame: render_vulkan.cpp:1937-1941
    if (reset_accumulation) {
        if (!h->freeze_frame) h->frame_offset += h->frame_id;
        h->frame_id = 0;
    }
    const uint32_t frame_id_before = h->frame_id;
    RpFrame f;
    fill_frame_constants(h, c, camera, per_frame_cameras, variant, spp, n_frames, reset_rest, f);
    size_t ev_cursor = 0;
    c.spans.clear();
    auto timed_on = [&](hipStream_t st, int kind, auto &&launch) {
        if (h->stage_timing >= 2 || (h->stage_timing == 1 && kind == 0)) {
            hipEvent_t a = next_event(c, ev_cursor), b = next_event(c, ev_cursor);
            (void)hipEventRecord(a, st);
            launch();
            (void)hipEventRecord(b, st);
            c.spans.push_back({a, b, kind});
        } else
            launch();
    };
    auto timed = [&](int kind, auto &&launch) { timed_on(c.stream, kind, launch); };
    // a stage that is ONE kernel: its start / stop events ride on the dispatch packet itself (hipExtLaunchKernelGGL), no extra
    // barrier packets in the queue -- the command processor's packet rate is what bounds small frames (profiles/r01_notes.md)
    auto timed_kernel = [&](hipStream_t st, int kind, auto kernel, dim3 grid, dim3 block, auto... args) {
        if (h->stage_timing >= 2 || (h->stage_timing == 1 && kind == 0)) {
            hipEvent_t a = next_event(c, ev_cursor), b = next_event(c, ev_cursor);
            hipExtLaunchKernelGGL(kernel, grid, block, 0, st, a, b, 0, args...);
            c.spans.push_back({a, b, kind});
        } else
            hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
    };
    // ... the same for the path stages, whose kernels are picked by the launchers of launch.h
    auto timed_launch = [&](hipStream_t st, int kind, unsigned grid) -> RpLaunch {
        RpLaunch l = {grid, st, nullptr, nullptr};
        if (h->stage_timing >= 2 || (h->stage_timing == 1 && kind == 0)) {
            l.start = next_event(c, ev_cursor);
            l.stop = next_event(c, ev_cursor);
            c.spans.push_back({l.start, l.stop, kind});
        }
        return l;
    };
    // the general instantiation of the path stages: a table point set, or a screen jitter (raster TAA) -- the shipped path carries neither
    const bool table_rng_later = h->rng_variant != RPTR_RNG_VARIANT_UNIFORM || h->params.enable_raster_taa != 0;
    const bool table_rng = table_rng_later;
    bool side = c.side != nullptr, alone = h->ctx.size() == 1;
    if (h->ctx.size() == 2) {
        alone = true;
        for (FrameCtx &o : h->ctx)
            if (&o != &c && o.pending && !o.synced && hipEventQuery(o.ev_end) != hipSuccess) alone = false; // another frame is in flight: it fills the GPU
    }
    if (side && h->side_only_alone && !alone) side = false;
    const bool full = alone && h->alone_blocks[0] > 0 && !count_traversal;
    const int blocks_first = full ? h->alone_blocks[0] : h->persistent_blocks, blocks_later = full ? h->alone_blocks[1] : h->extend_later_blocks;
    const int blocks_connect[2] = {full ? h->alone_blocks[2] : h->connect_blocks[0], full ? h->alone_blocks[3] : h->connect_blocks[1]};

    SceneCopy &scn = h->ctx_scene.empty() ? h->master : h->ctx_scene[(size_t)(&c - h->ctx.data())];
    const bool follow = !h->ctx_scene.empty() && scn.version != h->refit_version;
    if (follow) {
        // this context's own vertices follow the master set: the copy of the float positions is queued on the backend's stream,
        // behind the caller's updates (this context is idle, the others keep rendering from their own sets)
        for (size_t gi = 0; gi < scn.dynpos.size(); ++gi)
            if (scn.dynpos[gi])
                HIP_TRY(h, hipMemcpyAsync(scn.dynpos[gi], h->master.dynpos[gi], (size_t)h->geom_tris[gi] * 9 * sizeof(float), hipMemcpyDeviceToDevice,
                                          h->stream));
    }
    if (multi) { // whatever the caller queued on the backend's stream (vertex updates, the copy above) comes first
        HIP_TRY(h, hipEventRecord(c.ev_dep, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(c.stream, c.ev_dep, 0));
    }
    if (follow) { // ... and its tree is refitted on its OWN stream: the refits of different contexts run side by side
        int err = RPTR_OK;
        (void)refit_scene_copy(h, scn, true, c.stream, &err);
        scn.version = h->refit_version;
        // a rebuild that could not start (no memory for its work space): the tree was refitted on its old topology, so this context is
        // consistent and the frame is rendered on it; the next refit tries again. The caller can tell: rptr_hip_bvh_rebuild_count does not
        // advance, and the failures are counted (rptr_hip_get_option(h, "bvh_rebuild_failures"))
        if (err != RPTR_OK) h->rebuild_failures++;
    }
    if (c.gather_pending) { // the image this context produced last is still being sent to rank 0 (host_comm.h)
        HIP_TRY(h, hipStreamWaitEvent(c.stream, c.ev_gather, 0));
        c.gather_pending = false;
    }
    HIP_TRY(h, hipEventRecord(c.ev_begin, c.stream));
    c.launches_extend = c.launches_connect = 0;
    memset(&c.earlier_batches, 0, sizeof(c.earlier_batches));
    memset(c.host_counters, 0, sizeof(RpCounters));
    int remaining = spp * n_frames; // (n_frames > 1: one internal batch holds them all, checked above)
    const bool local_work = h->local_rows > 0;
    f.frame_id = h->frame_id; // the whole call is one frame of the reference (its batch_spp = spp), whatever the internal batches
    f.alpha_test = h->uses_alpha ? 1 : 0;
    f.rng_variant = h->rng_variant;
    f.rng_table = h->rng_table;
    const bool single = h->master.dscene.single_instance != 0;
    while (remaining > 0) {
        const int batch = std::min(remaining, h->max_batch_spp);
        f.sample_base = h->frame_id;
        f.batch_spp = batch;
        if (local_work) {
            HIP_TRY(h, hipMemsetAsync(c.counters, 0, sizeof(RpCounters), c.stream));
            // the first bounce's queue is the identity over the batch's path ids and is not stored (kernels.h)
            const uint32_t first_count = (uint32_t)((size_t)batch * h->npix_padded);
            const uint32_t *first_ids = nullptr;
            HIP_TRY(h, hipMemsetD32Async((hipDeviceptr_t)&c.counters->bounce[0].queue_count, (int)first_count, 1, c.stream));
            // the late bounces in one launch (kernels.h rp_k_tail); counting keeps the stand-alone kernels
            int tail_from = h->params.max_path_depth;
            if (h->tail_mode != 0 && !count_traversal)
                tail_from = std::max(1, std::min(h->params.max_path_depth, h->tail_mode > 0 ? h->tail_mode : h->tail_adaptive));
            c.tail_from = tail_from;
            for (int b = 0; b < h->params.max_path_depth; ++b) {
                const int in = b & 1, out = in ^ 1;
                RpBounceCounters *bc = &c.counters->bounce[b];
                if (b == tail_from) {
                    if (side && b > 0) HIP_TRY(h, hipStreamWaitEvent(c.stream, c.ev_side, 0)); // join: connect(b-1) on the side stream
                    const bool lights = (h->num_lights > 0 && !h->lights_disabled) || f.sp.sun_radiance[3] < 1.0f;
                    const bool full = h->uses_textures || h->uses_alpha; // one instantiation serves textured and alpha-tested scenes
                    rp_launch_tail(variant, h->opt.v[OPT_FAST_MATH] != 0, timed_launch(c.stream, 3, (unsigned)h->tail_blocks), lights, full, single, table_rng_later, scn.dscene, f, c.ps, c.sq,
                                   (const uint32_t *)c.queue[in], c.counters, b, c.gstack);
                    break;
                }
                rp_launch_extend(timed_launch(c.stream, 0, (unsigned)(b == 0 ? blocks_first : blocks_later)), count_traversal, b == 0, h->uses_alpha, single, b == 0 ? table_rng : table_rng_later, scn.dscene, f, c.ps,
                                 b == 0 ? first_ids : (const uint32_t *)c.queue[in], bc, c.counters, c.gstack);
                c.launches_extend++;
                const uint32_t *in_queue = b == 0 ? first_ids : c.queue[in];
                const uint32_t *order = in_queue;
                if (side && b > 0) HIP_TRY(h, hipStreamWaitEvent(c.stream, c.ev_side, 0)); // join: connect(b-1) wrote illum, frees the shadow queue
                timed(2, [&] {
                    launch_shade(h, c, variant, scn.dscene, f, order, b, out);
                });
                {
                    hipStream_t cs = side ? c.side : c.stream;
                    int *stack = side ? c.gstack_side : c.gstack;
                    if (side) { // fork: the side stream sees shade(b)
                        HIP_TRY(h, hipEventRecord(c.ev_fork, c.stream));
                        HIP_TRY(h, hipStreamWaitEvent(c.side, c.ev_fork, 0));
                    }
                    rp_launch_connect(timed_launch(cs, 1, (unsigned)blocks_connect[single ? 1 : 0]), count_traversal, h->uses_alpha, single, scn.dscene, f, c.ps, c.sq, bc, c.counters,
                                      stack);
                    if (side) HIP_TRY(h, hipEventRecord(c.ev_side, c.side));
                }
                c.launches_connect++;
            }
            if (side) HIP_TRY(h, hipStreamWaitEvent(c.stream, c.ev_side, 0)); // the last connect
            // resolves fold into one history buffer: they run in submission order across the contexts
            if (multi && h->last_resolved && h->last_resolved != c.ev_resolved) HIP_TRY(h, hipStreamWaitEvent(c.stream, h->last_resolved, 0));
            {
                const size_t npix = (size_t)h->width * h->local_rows;
                timed_kernel(c.stream, 4, rp_k_resolve, dim3(grid_for(h, npix)), dim3(256), f, c.ps, h->accum, h->fb, c.out_accum, c.out_fb);
            }
            if (multi) { // (the resolve also kept a copy of the image this frame produced: the next frame's resolve overwrites the shared buffers)
                HIP_TRY(h, hipEventRecord(c.ev_resolved, c.stream));
                h->last_resolved = c.ev_resolved;
            }
            HIP_TRY(h, hipMemcpyAsync(c.host_counters, c.counters, sizeof(RpCounters), hipMemcpyDeviceToHost, c.stream));
            // the host copy above must land before the next batch's memset: batches are few, sync here
            if (remaining - batch > 0) {
                HIP_TRY(h, hipStreamSynchronize(c.stream));
                add_counters(c.earlier_batches, *c.host_counters);
                memset(c.host_counters, 0, sizeof(RpCounters));
            }
        }
        // end_frame: render_vulkan.cpp:2152-2154
        if (n_frames == 1) {
            h->accumulated_spp = int(h->frame_id) + batch;
            h->frame_id += (uint32_t)batch;
        }
        remaining -= batch;
    }
    c.batch_spp_after[0] = h->accumulated_spp;
    if (n_frames > 1) { // begin_frame / end_frame of every frame of the batch (kernels: dshade.h rp_slot_frame)
        for (int k = 0; k < n_frames; ++k) {
            if (k > 0 && reset_rest) {
                h->frame_offset += h->frame_id;
                h->frame_id = 0;
            }
            h->frame_id += (uint32_t)spp;
            h->accumulated_spp = (int)h->frame_id;
            c.batch_spp_after[k] = h->accumulated_spp;
        }
    }
    HIP_TRY(h, hipEventRecord(c.ev_end, c.stream));
    HIP_TRY(h, hipGetLastError());
    if (h->freeze_frame) h->frame_id = frame_id_before; // end_frame, render_vulkan.cpp:2152-2154: the next frame repeats these samples
    c.spp_after = h->accumulated_spp;
    c.pending = true;
    c.synced = false;
    c.collected = 0;
    c.batch_n = n_frames;
    c.ticket = h->next_ticket;
    h->next_ticket += (uint64_t)n_frames;
    if (out_tickets)
        for (int k = 0; k < n_frames; ++k) out_tickets[k] = c.ticket + (uint64_t)k;
    return RPTR_OK;
}
} // extern "C++"

int rptr_hip_wait(rptr_hip_t *h, uint64_t ticket, RptrStats *out_stats) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    HIP_TRY(h, hipSetDevice(h->device));
    for (FrameCtx &c : h->ctx)
        if (c.pending && ticket >= c.ticket && ticket < c.ticket + (uint64_t)c.batch_n) {
            const int which = (int)(ticket - c.ticket);
            if (c.collected & (1u << which)) break; // waited for already
            return finish_frame(h, c, out_stats, which);
        }
    return fail(h, RPTR_E_INVALID, "ticket %llu is not in flight", (unsigned long long)ticket);
}

int rptr_hip_render(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp, int reset_accumulation, int count_traversal,
                    RptrStats *out_stats) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    int rc = drain(h); // a synchronous frame goes behind whatever is still in flight
    if (rc) return rc;
    uint64_t ticket = 0;
    if ((rc = rptr_hip_render_async(h, camera, variant, spp, reset_accumulation, count_traversal, &ticket))) return rc;
    return rptr_hip_wait(h, ticket, out_stats);
}

int rptr_hip_stats(const rptr_hip_t *h, RptrStats *out) {
    if (!h || !out) return fail(nullptr, RPTR_E_INVALID, "NULL argument");
    *out = h->stats;
    return RPTR_OK;
}

