// render_hip.hpp -- C++ host class over the C ABI, shaped like the reference's
// `struct RenderBackend` (librender/render_backend.h:68-116) so that the adapter a
// maintainer adds on the reference side is a thin subclass shim:
// same method names, same argument meaning, same error convention (failures throw
// a std::runtime_error like `logged_exception`, util/error_io.h:27-29; read-backs
// return the element count or 0, render_vulkan.cpp:2256-2275; configure_for
// returns bool). glm-free: vectors are float[3].
#pragma once
#include "../../include/rptr_hip.h"

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace rptr {

struct RenderStats { // librender/render_backend.h:15-24
    float render_time = 0;
    float rays_per_second = 0;
    int spp = 0;
    short frame_stats_delay = 0;
    bool has_valid_frame_stats = true;
    size_t total_device_bytes_allocated = 0;
};

struct RenderCameraParams { // librender/render_backend.h:26-31
    float pos[3], dir[3], up[3];
    float fovy;
};

struct RenderConfiguration { // librender/render_backend.h:33-40
    RenderCameraParams camera;
    double time = 0.0;
    int active_variant = 0;
    bool reset_accumulation = false;
    bool freeze_frame = false;
    int active_swap_buffer_count = -1; // :37: > 0 limits the frames in flight (the application sets 1 when it renders synchronously, app.cpp:386-389)
};

// util/display/render_graphic.h: `CommandStream*` is what the application hands to begin_frame / draw_frame / end_frame -- the display's
// stream (asynchronous: the frame is queued, up to MAX_SWAP_BUFFERS of them are in flight, statistics lag frame_stats_delay frames) or
// nullptr (synchronous, app.cpp:385-389). Here a frame's launches go to the HIP streams of the library's frame contexts, so the type only
// carries the distinction.
struct CommandStream {};

class RenderHip {
public:
    // public data members the app mutates directly (render_backend.h:69-76)
    RptrRenderParams params;
    RptrLightSamplingConfig lighting_params;
    RenderCameraParams camera;
    bool reset_accumulation = false;
    bool freeze_frame = false;

    static const int MAX_SWAP_BUFFERS = 2; // util/display/render_graphic.h:19

    // frames_in_flight: the library's frame contexts. The reference's backend owns MAX_SWAP_BUFFERS sets of per-frame resources whatever the
    // application does with them (vulkan/render_vulkan.h:128-131); so does this one by default: draw_frame(cmd_stream != nullptr) then has
    // two frames in flight, draw_frame(nullptr) renders one at a time on the same handle.
    // create_flags: RPTR_CREATE_*. 0 (an embedded plugin): the library touches nothing outside its handle; a host that owns its process
    // (bin/rptr_hip) passes RPTR_CREATE_SET_HW_QUEUES so that every frame context's stream gets a hardware queue (INTEGRATION.md "Hardware queues").
    explicit RenderHip(int device_ordinal = 0, int rank = 0, int world_size = 1, int stripe_rows = 32, void *hip_stream = nullptr,
                       int frames_in_flight = MAX_SWAP_BUFFERS, uint32_t create_flags = 0u) {
        RptrCreateInfo info{device_ordinal, rank, world_size, stripe_rows, hip_stream, frames_in_flight, RPTR_HIP_ABI_VERSION, create_flags, 0u};
        int rc = rptr_hip_create(&info, &h_);
        if (rc != RPTR_OK) throw std::runtime_error(std::string("rptr_hip_create: ") + rptr_hip_last_error(nullptr));
        contexts_ = frames_in_flight < 1 ? 1 : (frames_in_flight > 16 ? 16 : frames_in_flight);
        params = RptrRenderParams{1, RPTR_MAX_PATH_DEPTH, RPTR_DEFAULT_RR_PATH_DEPTH, 0, 0.f, 2.5f, 1.f, 4.f, 0, 0, 0.f, -1, 0, 8, 0, 1, 35.f, 0, 0, 0};
        lighting_params = RptrLightSamplingConfig{0.f, 16, 15.f, 0.f};
    }
    ~RenderHip() { rptr_hip_destroy(h_); }
    RenderHip(const RenderHip &) = delete;
    RenderHip &operator=(const RenderHip &) = delete;

    std::string name() const { return rptr_hip_name(); }
    std::vector<std::string> variant_names() const { return {"wavefront-gltf", "wavefront-diffuse", "wavefront-gltf-transmission"}; }

    // library options (include/rptr_hip.h "Options"): per-handle switches that take effect at initialize / set_scene / the next frame
    void set_option(const char *key, int64_t value) { check(rptr_hip_set_option(h_, key, value)); }
    int64_t get_option(const char *key) const {
        int64_t v = 0;
        check(rptr_hip_get_option(h_, key, &v));
        return v;
    }
    void initialize(const int fb_width, const int fb_height) {
        flush_pipeline();
        check(rptr_hip_initialize(h_, fb_width, fb_height));
    }
    // set_scene(const Scene&): the adapter flattens Scene into RptrSceneDesc
    void set_scene(const RptrSceneDesc &scene) {
        flush_pipeline();
        check(rptr_hip_set_scene(h_, &scene));
    }
    // update_config(SceneConfig): the adapter runs the reference's Hosek fit and passes the result
    void update_config(const RptrSceneParams &scene_params) { scene_params_ = scene_params; have_scene_params_ = true; }
    // dynamic meshes: float positions of one geometry (3 per unrolled vertex), then BLAS update + TLAS refit
    void update_vertices(uint32_t geometry, const float *xyz, uint32_t num_vertices) { check(rptr_hip_update_vertices(h_, geometry, xyz, num_vertices)); }
    void update_vertices_device(uint32_t geometry, const float *device_xyz, uint32_t num_vertices) {
        check(rptr_hip_update_vertices_device(h_, geometry, device_xyz, num_vertices));
    }
    void refit() { check(rptr_hip_refit(h_)); }
    // RenderBackendOptions (render_params.glsl.h:56-93): the point set with the table its render extension uploads, and the BVH policy
    void set_rng_variant(int rng_variant, const std::vector<uint32_t> &table = {}) {
        check(rptr_hip_set_rng_variant(h_, rng_variant, table.empty() ? nullptr : table.data(), table.size() * sizeof(uint32_t)));
    }
    void set_bvh_policy(bool force_bvh_rebuild, int rebuild_triangle_budget) {
        check(rptr_hip_set_bvh_policy(h_, force_bvh_rebuild ? 1 : 0, rebuild_triangle_budget));
    }
    bool configure_for(int variant_idx) {
        if (variant_idx != RPTR_VARIANT_GLTF && variant_idx != RPTR_VARIANT_SIMPLE && variant_idx != RPTR_VARIANT_GLTF_TRANSMISSION) return false;
        variant_ = variant_idx;
        return true;
    }

    // ---- the reference's frame loop (app.cpp:453-469): begin_frame -> draw_frame -> end_frame with the application's CommandStream*.
    // cmd_stream != nullptr: the frame is SUBMITTED (rptr_hip_render_async, this frame's camera) and the call returns; at most
    // active_swap_buffer_count (<= MAX_SWAP_BUFFERS <= the handle's frame contexts) frames are in flight: begin_frame of frame i + 2 waits for
    // frame i -- as RenderVulkan::begin_frame waits for the render-done event of the swap index it is about to reuse
    // (vulkan/render_vulkan.cpp:1953-1972) -- and takes frame i's timings, so stats() lags frame_stats_delay = 2 frames (:2229-2243).
    // cmd_stream == nullptr: synchronous -- everything in flight is finished first, stats() is this frame's.
    // Read-backs return the newest frame: they wait for everything in flight (the reference's read-back goes through the synchronous
    // command stream of the same queue, :2256-2275).
    void begin_frame(CommandStream *cmd_stream, const RenderConfiguration &config) {
        begin_frame(config);
        int active = config.active_swap_buffer_count > 0 ? config.active_swap_buffer_count : MAX_SWAP_BUFFERS;
        active = active < contexts_ ? active : contexts_;
        if (active > MAX_SWAP_BUFFERS) active = MAX_SWAP_BUFFERS;
        if (!cmd_stream || active != active_swap_) flush_pipeline(); // (a change of depth: start from an empty ring)
        active_swap_ = cmd_stream ? active : 1;
        swap_index_ = (swap_index_ + 1) % active_swap_;
        wait_slot(swap_index_); // the frame that used this swap index: done by now or waited for, its timings are the stats of this frame
    }
    void draw_frame(CommandStream *cmd_stream, int variant_idx) {
        if (variant_idx >= 0 && !configure_for(variant_idx)) throw std::runtime_error("rptr_hip: unknown variant");
        if (!cmd_stream) {
            draw_frame(0);
            asynchronous_ = false;
            return;
        }
        push_state();
        const RptrCamera cam = abi_camera();
        check(rptr_hip_render_async(h_, &cam, variant_, batch_spp(0), reset_accumulation ? 1 : 0, 0, &ticket_[swap_index_]));
        pending_[swap_index_] = true;
        asynchronous_ = true;
        count_frame(batch_spp(0));
    }
    void end_frame(CommandStream * /*cmd_stream*/, int /*variant_idx*/) {} // process_samples is sequenced inside the frame's own launch sequence
    void flush_pipeline() { // RenderBackend::flush_pipeline (vulkan/render_vulkan.cpp:2245-2248): nothing in flight afterwards
        for (int k = 1; k <= MAX_SWAP_BUFFERS; ++k) { // oldest first: the newest frame is the one waited for last (what read-backs return)
            wait_slot((swap_index_ + k) % MAX_SWAP_BUFFERS);
        }
    }

    void begin_frame(const RenderConfiguration &config) { // render_backend.cpp:17-23
        camera = config.camera;
        reset_accumulation = config.reset_accumulation;
        freeze_frame = config.freeze_frame;
        configure_for(config.active_variant);
    }
    void draw_frame(int spp = 0) { // synchronous
        flush_pipeline();
        push_state();
        const RptrCamera cam = abi_camera();
        check(rptr_hip_render(h_, &cam, variant_, batch_spp(spp), reset_accumulation ? 1 : 0, 0, &last_));
        ++stats_serial_;
        count_frame(batch_spp(spp));
        asynchronous_ = false;
    }
    // frames in flight: begin_frame + asynchronous draw_frame, collected with wait(ticket)
    // ---- explicit tickets (hosts that schedule deeper than the reference's two swap buffers: render_group.hpp, bin/rptr_hip --frames-in-flight)
    uint64_t render_async(const RenderConfiguration &config, int spp = 0) {
        begin_frame(config);
        push_state();
        const RptrCamera cam = abi_camera();
        uint64_t ticket = 0;
        check(rptr_hip_render_async(h_, &cam, variant_, batch_spp(spp), reset_accumulation ? 1 : 0, 0, &ticket));
        count_frame(batch_spp(spp));
        return ticket;
    }
    // several frames of the same view in ONE launch sequence (rptr_hip_render_batch_async): tickets[k] is frame k's
    std::vector<uint64_t> render_batch_async(const RenderConfiguration &config, int spp, int n_frames, bool reset_rest = true) {
        begin_frame(config);
        push_state();
        const RptrCamera cam = abi_camera();
        std::vector<uint64_t> tickets((size_t)n_frames, 0);
        check(rptr_hip_render_batch_async(h_, &cam, variant_, batch_spp(spp), n_frames, reset_accumulation ? 1 : 0, reset_rest ? 1 : 0, 0, tickets.data()));
        for (int k = 0; k < n_frames; ++k) {
            count_frame(batch_spp(spp));
            if (k + 1 < n_frames) reset_accumulation = reset_rest;
        }
        reset_accumulation = false;
        return tickets;
    }
    // ... with a camera per frame (rptr_hip_render_batch_cameras_async): the application moves the camera every frame (app.cpp:350-469);
    // configs[k] is frame k's RenderConfiguration (camera; variant / freeze of configs[0]; reset_accumulation of configs[0] for frame 0,
    // reset_rest for the others -- a moved camera restarts the accumulation)
    std::vector<uint64_t> render_batch_cameras_async(const RenderConfiguration *configs, int n_frames, int spp = 0, bool reset_rest = true) {
        begin_frame(configs[0]);
        push_state();
        std::vector<RptrCamera> cams((size_t)n_frames);
        for (int k = 0; k < n_frames; ++k) {
            camera = configs[k].camera;
            cams[(size_t)k] = abi_camera();
        }
        std::vector<uint64_t> tickets((size_t)n_frames, 0);
        check(rptr_hip_render_batch_cameras_async(h_, cams.data(), variant_, batch_spp(spp), n_frames, reset_accumulation ? 1 : 0, reset_rest ? 1 : 0, 0, tickets.data()));
        for (int k = 0; k < n_frames; ++k) {
            count_frame(batch_spp(spp));
            if (k + 1 < n_frames) reset_accumulation = reset_rest;
        }
        reset_accumulation = false;
        return tickets;
    }
    RenderStats wait(uint64_t ticket) {
        check(rptr_hip_wait(h_, ticket, &last_));
        ++stats_serial_;
        RenderStats s = stats();
        s.spp = last_.spp; // (the waited frame's own count)
        s.frame_stats_delay = 0;
        return s;
    }
    void end_frame() {} // process_samples is sequenced inside draw_frame on the same stream
    RenderStats render(const RenderConfiguration &config, int spp = 0) {
        begin_frame(config);
        draw_frame(spp);
        end_frame();
        return stats();
    }
    // render_vulkan.cpp:2229-2243: the timings of the frame whose swap buffers this frame took over (asynchronous: frame_stats_delay frames
    // ago), spp = samples accumulated INCLUDING the frame just submitted (:2152-2154). rays_per_second is filled here; the reference leaves -1
    RenderStats stats() const {
        RenderStats s;
        s.render_time = last_.render_time_ms;
        s.has_valid_frame_stats = last_.render_time_ms != 0.0f;
        s.rays_per_second = s.has_valid_frame_stats ? float(double(last_.rays_closest + last_.rays_shadow) / (last_.render_time_ms * 1e-3)) : -1.f;
        s.frame_stats_delay = short(asynchronous_ ? active_swap_ : 0);
        s.spp = accumulated_spp_;
        s.total_device_bytes_allocated = (size_t)last_.device_bytes_allocated;
        return s;
    }
    uint64_t rays_of_last_stats() const { return last_.rays_closest + last_.rays_shadow; }
    uint64_t stats_serial() const { return stats_serial_; } // grows whenever stats() starts to describe another frame

    void get_framebuffer_size(uint32_t whc[3]) const { check(rptr_hip_get_framebuffer_size(h_, whc)); }
    size_t readback_framebuffer(size_t buffer_size, float *buffer) { // RGBA32F accumulation buffer
        uint32_t whc[3];
        get_framebuffer_size(whc);
        const size_t need = size_t(whc[0]) * whc[1] * 4;
        if (buffer_size < need) return 0;
        flush_pipeline();
        check(rptr_hip_readback_f32(h_, buffer, buffer_size));
        return need;
    }
    size_t readback_framebuffer(size_t buffer_size, unsigned char *buffer) { // sRGB RGBA8
        uint32_t whc[3];
        get_framebuffer_size(whc);
        const size_t need = size_t(whc[0]) * whc[1] * 4;
        if (buffer_size < need) return 0;
        flush_pipeline();
        check(rptr_hip_readback_u8(h_, buffer, buffer_size));
        return need;
    }

    // RenderGraphic::readback_aov (util/display/render_graphic.h:12-17,40): RGBA16F, the reference's AOVBufferIndex order
    enum AOVBufferIndex { AOVAlbedoRoughnessIndex = 0, AOVNormalDepthIndex, AOVMotionJitterIndex, AOVBufferCount };
    size_t readback_aov(AOVBufferIndex aov_index, size_t buffer_size, uint16_t *buffer, bool /*force_refresh*/ = false) {
        uint32_t whc[3];
        get_framebuffer_size(whc);
        const size_t need = size_t(whc[0]) * whc[1] * 4;
        if (buffer_size < need) return 0;
        flush_pipeline();
        check(rptr_hip_readback_aov(h_, int(aov_index), buffer, buffer_size));
        return need;
    }

    // enable_ray_queries / render_ray_queries with RQ_CLOSEST (render_backend.h:101-102; vulkan/render_vulkan.cpp:430-455,1867-1876): the
    // queries live in a DEVICE buffer the backend owns (ray_query_buffer), other device code fills it, the results land in
    // ray_result_buffer. ray_query_buffer() / ray_result_buffer() are the device addresses.
    void enable_ray_queries(const int max_queries, const int max_queries_per_pixel = 0) {
        check(rptr_hip_enable_ray_queries(h_, max_queries, max_queries_per_pixel, &ray_query_buffer_, &ray_result_buffer_));
    }
    void *ray_query_buffer() const { return ray_query_buffer_; }
    void *ray_result_buffer() const { return ray_result_buffer_; }
    bool render_ray_queries(int num_queries) { // (RenderParams / variant / command stream of the reference's signature select nothing here)
        if (!ray_query_buffer_) return false;
        check(rptr_hip_render_ray_queries(h_, num_queries));
        return true;
    }
    // convenience over HOST arrays (tests, tools): rptr_hip_trace uploads, traces, reads back
    bool render_ray_queries(const RptrRenderRayQuery *queries, int num_queries, float *results4) {
        if (num_queries < 0) return false;
        // (rptr_hip_trace sizes its own staging: the budget of enable_ray_queries only bounds the device buffers)
        check(rptr_hip_trace(h_, queries, num_queries, results4));
        return true;
    }
    // RenderBackendOptions::light_sampling_variant (rendering/mc/light_sampling.h:11-20): 0 NONE, 1 RIS; light_sampling_bucket_count is
    // LightSamplingConfig::bin_size, at most RPTR_BINNED_LIGHTS_BIN_MAX_SIZE (set_params refuses more)
    void set_light_sampling_variant(int variant) { check(rptr_hip_set_light_sampling_variant(h_, variant)); }
    rptr_hip_t *handle() { return h_; }

private:
    void check(int rc) const {
        if (rc != RPTR_OK) throw std::runtime_error(std::string("rptr_hip: ") + rptr_hip_last_error(h_));
    }
    void wait_slot(int i) {
        if (!pending_[i]) return;
        pending_[i] = false;
        // (a library call that needs the device to itself -- set_scene, ray queries over host arrays, a vertex update of a scene without
        // per-context copies -- has finished every frame in flight already: such a ticket is "not in flight" any more, its timings are gone)
        const int rc = rptr_hip_wait(h_, ticket_[i], &last_);
        if (rc == RPTR_OK) ++stats_serial_;
        if (rc != RPTR_OK && !(rc == RPTR_E_INVALID && std::string(rptr_hip_last_error(h_)).find("not in flight") != std::string::npos)) check(rc);
    }
    int batch_spp(int spp) const { return spp > 0 ? spp : (params.batch_spp > 0 ? params.batch_spp : 1); }
    void push_state() { // RenderBackend::begin_frame "update params" (render_backend.cpp:17-23): the public members as they are NOW
        check(rptr_hip_set_params(h_, &params, have_scene_params_ ? &scene_params_ : nullptr, &lighting_params));
        check(rptr_hip_set_freeze_frame(h_, freeze_frame ? 1 : 0));
    }
    RptrCamera abi_camera() const {
        RptrCamera cam;
        for (int k = 0; k < 3; ++k) {
            cam.pos[k] = camera.pos[k];
            cam.dir[k] = camera.dir[k];
            cam.up[k] = camera.up[k];
        }
        cam.fovy = camera.fovy;
        return cam;
    }
    void count_frame(int spp) { // begin_frame / end_frame bookkeeping (render_vulkan.cpp:1937-1941,2152-2154), as the library keeps it
        if (reset_accumulation) frame_id_ = 0;
        accumulated_spp_ = int(frame_id_) + spp;
        if (!freeze_frame) frame_id_ += (uint32_t)spp;
        reset_accumulation = false;
    }
    rptr_hip_t *h_ = nullptr;
    int contexts_ = 1, active_swap_ = 1, swap_index_ = 0;
    uint64_t ticket_[MAX_SWAP_BUFFERS] = {0, 0};
    bool pending_[MAX_SWAP_BUFFERS] = {false, false};
    bool asynchronous_ = false;
    uint32_t frame_id_ = 0;
    uint64_t stats_serial_ = 0;
    int accumulated_spp_ = 0;
    RptrSceneParams scene_params_{};
    bool have_scene_params_ = false;
    int variant_ = RPTR_VARIANT_GLTF;
    void *ray_query_buffer_ = nullptr, *ray_result_buffer_ = nullptr;
    RptrStats last_{};
};

// ≙ struct RaytraceBackend (librender/raytrace_backend.h:13-19; libdatacapture's trace_ray over host arrays): closest-hit queries through
// rptr_hip_trace. The adapter on the reference's side converts rt_datacapture::RayQuery {origin, dir, t_max} to RptrRenderRayQuery
// (mode_or_data = 0) and RaytraceResults from the float4 rows (barycentrics, instance + geometry index, primitive index).
class RaytraceHip {
public:
    explicit RaytraceHip(int device_ordinal = 0) : backend_(device_ordinal, 0, 1, 32, nullptr, 1) { backend_.initialize(8, 8); } // (queries need no frame, the stack scratch does)
    std::string name() const { return std::string(rptr_hip_name()) + " (ray queries)"; }
    void set_scene(const RptrSceneDesc &scene) { backend_.set_scene(scene); }
    // returns the number of queries traced; results4: 4 floats per query in rt_intersect.comp's layout, miss = (-1, -1, bits(-1), bits(-1))
    int trace_ray(const RptrRenderRayQuery *queries, int num_queries, float *results4) {
        return backend_.render_ray_queries(queries, num_queries, results4) ? num_queries : 0;
    }
    RenderHip &backend() { return backend_; }

private:
    RenderHip backend_;
};

} // namespace rptr
