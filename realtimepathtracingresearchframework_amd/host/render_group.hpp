// render_group.hpp -- one host process driving several GPUs: N `RenderHip` backends (one per device, rank i of N) that share a frame
// by screen stripes, plus the library's gather of tile radiance to rank 0 (include/rptr_hip.h "multi-GPU": rptr_hip_comm_init_all /
// rptr_hip_gather_all -- grouped ncclSend / ncclRecv over xGMI, or peer copies when handles share a device).
// Shaped like one RenderBackend: the application calls set_scene / render / readback once, the group fans out. No reference
// counterpart (the reference renders on one physical device, vulkan/render_vulkan_extensions.cpp:77-82); partitioning per SURVEY 8e.
#pragma once
#include "render_hip.hpp"

#include <memory>
#include <vector>

namespace rptr {

class RenderGroup {
public:
    // devices: HIP ordinals, one rank each (the same ordinal may appear several times: a test rig on one GPU)
    RenderGroup(const std::vector<int> &devices, int stripe_rows = 8, int frames_in_flight = RenderHip::MAX_SWAP_BUFFERS, uint32_t create_flags = 0u) {
        const int n = (int)devices.size();
        if (n < 1) throw std::runtime_error("RenderGroup: no devices");
        for (int i = 0; i < n; ++i) ranks_.emplace_back(new RenderHip(devices[(size_t)i], i, n, stripe_rows, nullptr, frames_in_flight, create_flags));
    }
    int size() const { return (int)ranks_.size(); }
    RenderHip &rank(int i) { return *ranks_[(size_t)i]; }
    std::string name() const { return ranks_[0]->name() + " x" + std::to_string(size()); }

    void initialize(int fb_width, int fb_height) {
        width_ = fb_width;
        height_ = fb_height;
        for (auto &r : ranks_) r->initialize(fb_width, fb_height);
        if (size() > 1) { // (a communicator's buffers are frame-sized: it is made again after every initialize)
            std::vector<rptr_hip_t *> hs;
            for (auto &r : ranks_) hs.push_back(r->handle());
            if (rptr_hip_comm_init_all(hs.data(), size()) != RPTR_OK) throw std::runtime_error(std::string("rptr_hip_comm_init_all: ") + last_error());
        }
    }
    void set_scene(const RptrSceneDesc &scene) { // every GPU holds a replica of the scene
        for (auto &r : ranks_) r->set_scene(scene);
    }
    void update_config(const RptrSceneParams &sp) {
        for (auto &r : ranks_) r->update_config(sp);
    }
    void set_params(const RptrRenderParams &p, const RptrLightSamplingConfig &l) {
        for (auto &r : ranks_) {
            r->params = p;
            r->lighting_params = l;
        }
    }
    void update_vertices(uint32_t geometry, const float *xyz, uint32_t num_vertices) {
        for (auto &r : ranks_) r->update_vertices(geometry, xyz, num_vertices);
    }
    void refit() {
        for (auto &r : ranks_) r->refit();
    }
    void set_rng_variant(int rng_variant, const std::vector<uint32_t> &table = {}) {
        for (auto &r : ranks_) r->set_rng_variant(rng_variant, table);
    }
    void set_bvh_policy(bool force_bvh_rebuild, int rebuild_triangle_budget) {
        for (auto &r : ranks_) r->set_bvh_policy(force_bvh_rebuild, rebuild_triangle_budget);
    }
    void set_option(const char *key, int64_t value) { // before initialize / set_scene (include/rptr_hip.h "Options")
        for (auto &r : ranks_) r->set_option(key, value);
    }
    int64_t get_option(const char *key) const { // the smallest value over the ranks (limits: "sample_slots")
        int64_t v = ranks_[0]->get_option(key);
        for (auto &r : ranks_) v = std::min(v, r->get_option(key));
        return v;
    }
    // The reference's frame loop (app.cpp:453-469) on ONE device: begin_frame / draw_frame / end_frame with the application's CommandStream*
    // (render_hip.hpp: non-null = submitted, two frames in flight, statistics two frames late). A group of several devices renders such a
    // frame synchronously (its gather follows the frame).
    RenderStats frame(CommandStream *cmd_stream, const RenderConfiguration &config) {
        if (size() > 1 || !cmd_stream) return render(config);
        RenderHip &r = *ranks_[0];
        r.begin_frame(cmd_stream, config);
        r.draw_frame(cmd_stream, config.active_variant);
        r.end_frame(cmd_stream, config.active_variant);
        const RenderStats s = r.stats();
        if (s.has_valid_frame_stats && r.stats_serial() != counted_serial_) { // (the timings of the frame two submissions ago: counted once)
            rays_ += double(r.rays_of_last_stats());
            counted_serial_ = r.stats_serial();
        }
        return s;
    }
    void flush_pipeline() {
        for (auto &r : ranks_) r->flush_pipeline();
    }
    // one frame on all GPUs: every rank renders its stripes (asynchronously, side by side), then the tiles are gathered to rank 0
    RenderStats render(const RenderConfiguration &config, int spp = 0) {
        std::vector<uint64_t> tickets;
        for (auto &r : ranks_) tickets.push_back(r->render_async(config, spp));
        RenderStats total{};
        for (size_t i = 0; i < ranks_.size(); ++i) {
            const RenderStats s = ranks_[i]->wait(tickets[i]);
            total.render_time = std::max(total.render_time, s.render_time);
            rays_ += s.has_valid_frame_stats ? double(s.rays_per_second) * s.render_time * 1e-3 : 0.0;
            total.spp = s.spp;
            total.total_device_bytes_allocated += s.total_device_bytes_allocated;
        }
        if (size() > 1) {
            std::vector<rptr_hip_t *> hs;
            for (auto &r : ranks_) hs.push_back(r->handle());
            if (rptr_hip_gather_all(hs.data(), size()) != RPTR_OK) throw std::runtime_error(std::string("rptr_hip_gather_all: ") + last_error());
        }
        return total;
    }
    // The pipelined schedule (what bench.py measures): a launch sequence of `n_frames` frames is queued on every GPU and collected later,
    // while others are in flight (the group's frames_in_flight contexts). collect() waits for the sequence's frames in order, gathers them
    // to rank 0 in ONE collective (rptr_hip_gather_all_batch: the frames of a sequence finish together and lie behind each other in every
    // rank's images; per_frame_gather = true: one collective per frame, as rounds 2-3 did) and returns one RenderStats per frame
    // (render_time: the slowest rank's share of the sequence). readback_framebuffer(.., frame) then reads frame k of that sequence.
    struct Sequence {
        std::vector<std::vector<uint64_t>> tickets; // [rank][frame]
        int frames = 0;
    };
    Sequence submit(const RenderConfiguration &config, int spp, int n_frames, bool reset_rest = true) {
        Sequence q;
        q.frames = n_frames;
        for (auto &r : ranks_) q.tickets.push_back(n_frames > 1 ? r->render_batch_async(config, spp, n_frames, reset_rest) : std::vector<uint64_t>{r->render_async(config, spp)});
        return q;
    }
    // ... with a camera per frame: configs[k] is frame k's (rptr_hip_render_batch_cameras_async; the application's loop may move the camera
    // every frame, app.cpp:350-469)
    Sequence submit(const std::vector<RenderConfiguration> &configs, int spp, bool reset_rest = true) {
        Sequence q;
        q.frames = (int)configs.size();
        for (auto &r : ranks_)
            q.tickets.push_back(q.frames > 1 ? r->render_batch_cameras_async(configs.data(), q.frames, spp, reset_rest) : std::vector<uint64_t>{r->render_async(configs[0], spp)});
        return q;
    }
    // one frame of a sequence at a time (hosts that act between the frames: an image per keyframe): the tickets of frame k on every rank,
    // the gather when it is the sequence's last (all its frames in one collective)
    RenderStats collect_frame(const Sequence &q, int k) {
        RenderStats total{};
        total.has_valid_frame_stats = false;
        for (size_t i = 0; i < ranks_.size(); ++i) {
            const RenderStats s = ranks_[i]->wait(q.tickets[i][(size_t)k]);
            total.render_time = std::max(total.render_time, s.render_time);
            total.has_valid_frame_stats = total.has_valid_frame_stats || s.has_valid_frame_stats;
            rays_ += s.has_valid_frame_stats ? double(s.rays_per_second) * s.render_time * 1e-3 : 0.0;
            total.spp = s.spp;
            total.total_device_bytes_allocated += s.total_device_bytes_allocated;
        }
        if (size() > 1 && k == q.frames - 1) {
            std::vector<rptr_hip_t *> hs;
            for (auto &r : ranks_) hs.push_back(r->handle());
            if (rptr_hip_gather_all_batch(hs.data(), size(), q.frames) != RPTR_OK) throw std::runtime_error(std::string("rptr_hip_gather_all_batch: ") + last_error());
        }
        return total;
    }
    std::vector<RenderStats> collect(const Sequence &q, bool per_frame_gather = false) {
        std::vector<RenderStats> out;
        for (int k = 0; k < q.frames; ++k) {
            RenderStats total{};
            for (size_t i = 0; i < ranks_.size(); ++i) {
                const RenderStats s = ranks_[i]->wait(q.tickets[i][(size_t)k]);
                total.render_time = std::max(total.render_time, s.render_time);
                total.has_valid_frame_stats = total.has_valid_frame_stats || s.has_valid_frame_stats;
                rays_ += s.has_valid_frame_stats ? double(s.rays_per_second) * s.render_time * 1e-3 : 0.0;
                total.spp = s.spp;
                total.total_device_bytes_allocated += s.total_device_bytes_allocated;
            }
            if (size() > 1 && (per_frame_gather || k == q.frames - 1)) {
                std::vector<rptr_hip_t *> hs;
                for (auto &r : ranks_) hs.push_back(r->handle());
                if (rptr_hip_gather_all_batch(hs.data(), size(), per_frame_gather ? 1 : q.frames) != RPTR_OK)
                    throw std::runtime_error(std::string("rptr_hip_gather_all_batch: ") + last_error());
            }
            out.push_back(total);
        }
        return out;
    }
    // frame k (0 = the oldest) of the sequence collected last, assembled on rank 0
    size_t readback_framebuffer(size_t buffer_size, float *buffer, int frame) {
        if (size() == 1) return 0; // (one rank: the frames of a sequence are read through the rank's own tickets)
        const size_t need = (size_t)width_ * height_ * 4;
        if (buffer_size < need) return 0;
        if (rptr_hip_readback_gathered_frame_f32(ranks_[0]->handle(), frame, buffer, buffer_size) != RPTR_OK) throw std::runtime_error(std::string("readback: ") + last_error());
        return need;
    }
    double rays_traced() const { return rays_; }
    // the full frame (RGBA32F accumulation buffer) on the host: rank 0's assembled frame
    size_t readback_framebuffer(size_t buffer_size, float *buffer) {
        if (size() == 1) return ranks_[0]->readback_framebuffer(buffer_size, buffer);
        const size_t need = (size_t)width_ * height_ * 4;
        if (buffer_size < need) return 0;
        if (rptr_hip_readback_gathered_f32(ranks_[0]->handle(), buffer, buffer_size) != RPTR_OK) throw std::runtime_error(std::string("readback: ") + last_error());
        return need;
    }
    // 8-bit frame buffer / AOV images: every rank fills in its own rows (the read-backs leave the other rows untouched)
    size_t readback_framebuffer(size_t buffer_size, unsigned char *buffer) {
        size_t n = 0;
        for (auto &r : ranks_) n = r->readback_framebuffer(buffer_size, buffer);
        return n;
    }
    size_t readback_aov(RenderHip::AOVBufferIndex aov, size_t buffer_size, uint16_t *buffer) {
        size_t n = 0;
        for (auto &r : ranks_) n = r->readback_aov(aov, buffer_size, buffer);
        return n;
    }
    float mean_gather_ms() {
        uint64_t n = 0;
        float ms = 0.f;
        if (size() > 1) (void)rptr_hip_comm_stats(ranks_[0]->handle(), &n, &ms);
        return ms;
    }

private:
    std::string last_error() {
        for (auto &r : ranks_) {
            const char *e = rptr_hip_last_error(r->handle());
            if (e && *e) return e;
        }
        return rptr_hip_last_error(nullptr);
    }
    std::vector<std::unique_ptr<RenderHip>> ranks_;
    int width_ = 0, height_ = 0;
    double rays_ = 0.0;
    uint64_t counted_serial_ = 0;
};

} // namespace rptr
