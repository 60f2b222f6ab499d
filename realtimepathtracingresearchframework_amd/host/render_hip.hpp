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
};

class RenderHip {
public:
    // public data members the app mutates directly (render_backend.h:69-76)
    RptrRenderParams params;
    RptrLightSamplingConfig lighting_params;
    RenderCameraParams camera;
    bool reset_accumulation = false;
    bool freeze_frame = false;

    explicit RenderHip(int device_ordinal = 0, int rank = 0, int world_size = 1, int stripe_rows = 32, void *hip_stream = nullptr,
                       int frames_in_flight = 1) {
        RptrCreateInfo info{device_ordinal, rank, world_size, stripe_rows, hip_stream, frames_in_flight, RPTR_HIP_ABI_VERSION};
        int rc = rptr_hip_create(&info, &h_);
        if (rc != RPTR_OK) throw std::runtime_error(std::string("rptr_hip_create: ") + rptr_hip_last_error(nullptr));
        params = RptrRenderParams{1, RPTR_MAX_PATH_DEPTH, RPTR_DEFAULT_RR_PATH_DEPTH, 0, 0.f, 2.5f, 1.f, 4.f, 0, 0, 0.f, -1, 0, 8, 0, 1, 35.f, 0, 0, 0};
        lighting_params = RptrLightSamplingConfig{0.f, 16, 15.f, 0.f};
    }
    ~RenderHip() { rptr_hip_destroy(h_); }
    RenderHip(const RenderHip &) = delete;
    RenderHip &operator=(const RenderHip &) = delete;

    std::string name() const { return rptr_hip_name(); }
    std::vector<std::string> variant_names() const { return {"wavefront-gltf", "wavefront-diffuse", "wavefront-gltf-transmission"}; }

    void initialize(const int fb_width, const int fb_height) { check(rptr_hip_initialize(h_, fb_width, fb_height)); }
    // set_scene(const Scene&): the adapter flattens Scene into RptrSceneDesc
    void set_scene(const RptrSceneDesc &scene) { check(rptr_hip_set_scene(h_, &scene)); }
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

    void begin_frame(const RenderConfiguration &config) { // render_backend.cpp:17-23
        camera = config.camera;
        reset_accumulation = config.reset_accumulation;
        freeze_frame = config.freeze_frame;
        configure_for(config.active_variant);
    }
    void draw_frame(int spp = 0) {
        check(rptr_hip_set_params(h_, &params, have_scene_params_ ? &scene_params_ : nullptr, &lighting_params));
        check(rptr_hip_set_freeze_frame(h_, freeze_frame ? 1 : 0));
        RptrCamera cam;
        for (int k = 0; k < 3; ++k) {
            cam.pos[k] = camera.pos[k];
            cam.dir[k] = camera.dir[k];
            cam.up[k] = camera.up[k];
        }
        cam.fovy = camera.fovy;
        check(rptr_hip_render(h_, &cam, variant_, spp > 0 ? spp : (params.batch_spp > 0 ? params.batch_spp : 1), reset_accumulation ? 1 : 0, 0,
                              &last_));
        reset_accumulation = false;
    }
    // frames in flight: begin_frame + asynchronous draw_frame, collected with wait(ticket)
    uint64_t render_async(const RenderConfiguration &config, int spp = 0) {
        begin_frame(config);
        check(rptr_hip_set_params(h_, &params, have_scene_params_ ? &scene_params_ : nullptr, &lighting_params));
        check(rptr_hip_set_freeze_frame(h_, freeze_frame ? 1 : 0));
        RptrCamera cam;
        for (int k = 0; k < 3; ++k) {
            cam.pos[k] = camera.pos[k];
            cam.dir[k] = camera.dir[k];
            cam.up[k] = camera.up[k];
        }
        cam.fovy = camera.fovy;
        uint64_t ticket = 0;
        check(rptr_hip_render_async(h_, &cam, variant_, spp > 0 ? spp : (params.batch_spp > 0 ? params.batch_spp : 1), reset_accumulation ? 1 : 0, 0,
                                    &ticket));
        reset_accumulation = false;
        return ticket;
    }
    // several frames of the same view in ONE launch sequence (rptr_hip_render_batch_async): tickets[k] is frame k's
    std::vector<uint64_t> render_batch_async(const RenderConfiguration &config, int spp, int n_frames, bool reset_rest = true) {
        begin_frame(config);
        check(rptr_hip_set_params(h_, &params, have_scene_params_ ? &scene_params_ : nullptr, &lighting_params));
        check(rptr_hip_set_freeze_frame(h_, freeze_frame ? 1 : 0));
        RptrCamera cam;
        for (int k = 0; k < 3; ++k) {
            cam.pos[k] = camera.pos[k];
            cam.dir[k] = camera.dir[k];
            cam.up[k] = camera.up[k];
        }
        cam.fovy = camera.fovy;
        std::vector<uint64_t> tickets((size_t)n_frames, 0);
        check(rptr_hip_render_batch_async(h_, &cam, variant_, spp > 0 ? spp : (params.batch_spp > 0 ? params.batch_spp : 1), n_frames, reset_accumulation ? 1 : 0,
                                          reset_rest ? 1 : 0, 0, tickets.data()));
        reset_accumulation = false;
        return tickets;
    }
    RenderStats wait(uint64_t ticket) {
        check(rptr_hip_wait(h_, ticket, &last_));
        return stats();
    }
    void end_frame() {} // process_samples is sequenced inside draw_frame on the same stream
    RenderStats render(const RenderConfiguration &config, int spp = 0) {
        begin_frame(config);
        draw_frame(spp);
        end_frame();
        return stats();
    }
    RenderStats stats() const { // render_vulkan.cpp:2229-2243 (rays_per_second is filled here; the reference leaves -1)
        RenderStats s;
        s.render_time = last_.render_time_ms;
        s.has_valid_frame_stats = last_.render_time_ms != 0.0f;
        s.rays_per_second = s.has_valid_frame_stats ? float(double(last_.rays_closest + last_.rays_shadow) / (last_.render_time_ms * 1e-3)) : -1.f;
        s.spp = last_.spp;
        s.total_device_bytes_allocated = (size_t)last_.device_bytes_allocated;
        return s;
    }

    void get_framebuffer_size(uint32_t whc[3]) const { check(rptr_hip_get_framebuffer_size(h_, whc)); }
    size_t readback_framebuffer(size_t buffer_size, float *buffer) { // RGBA32F accumulation buffer
        uint32_t whc[3];
        get_framebuffer_size(whc);
        const size_t need = size_t(whc[0]) * whc[1] * 4;
        if (buffer_size < need) return 0;
        check(rptr_hip_readback_f32(h_, buffer, buffer_size));
        return need;
    }
    size_t readback_framebuffer(size_t buffer_size, unsigned char *buffer) { // sRGB RGBA8
        uint32_t whc[3];
        get_framebuffer_size(whc);
        const size_t need = size_t(whc[0]) * whc[1] * 4;
        if (buffer_size < need) return 0;
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
        check(rptr_hip_readback_aov(h_, int(aov_index), buffer, buffer_size));
        return need;
    }

    // enable_ray_queries / render_ray_queries with RQ_CLOSEST (render_backend.h:101-102; vulkan/render_vulkan.cpp:430-455,1867-1876): the
    // queries live in a DEVICE buffer the backend owns (ray_query_buffer), other device code fills it, the results land in
    // ray_result_buffer. ray_query_buffer() / ray_result_buffer() are the device addresses.
    void enable_ray_queries(const int max_queries, const int max_queries_per_pixel = 0) {
        check(rptr_hip_enable_ray_queries(h_, max_queries, max_queries_per_pixel, &ray_query_buffer_, &ray_result_buffer_));
        max_queries_ = max_queries;
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
        if (num_queries > max_queries_) return false;
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
    rptr_hip_t *h_ = nullptr;
    RptrSceneParams scene_params_{};
    bool have_scene_params_ = false;
    int variant_ = RPTR_VARIANT_GLTF;
    int max_queries_ = 512 * 512;
    void *ray_query_buffer_ = nullptr, *ray_result_buffer_ = nullptr;
    RptrStats last_{};
};

// ≙ struct RaytraceBackend (librender/raytrace_backend.h:13-19; libdatacapture's trace_ray over host arrays): closest-hit queries through
// rptr_hip_trace. The adapter on the reference's side converts rt_datacapture::RayQuery {origin, dir, t_max} to RptrRenderRayQuery
// (mode_or_data = 0) and RaytraceResults from the float4 rows (barycentrics, instance + geometry index, primitive index).
class RaytraceHip {
public:
    explicit RaytraceHip(int device_ordinal = 0) : backend_(device_ordinal) { backend_.initialize(8, 8); } // (queries need no frame, the stack scratch does)
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
