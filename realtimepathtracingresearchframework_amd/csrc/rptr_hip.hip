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

namespace {

thread_local std::string g_last_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct MeshRt { // one bottom-level structure
    int node_base = 0;  // absolute index of the root in the shared node array
    int node_count = 0;
    int node_capacity = 0; // dynamic meshes reserve one node per triangle: what a device-side rebuild (lbvh.h) can need
    int tri_base = 0;
    int tri_count = 0;
    float lo[3], hi[3];
    bool dynamic = false;
    bool rebuildable = false; // dynamic and not RPTR_MESH_SUBTLY_DYNAMIC: the BVH policy may give it a new tree
};

// The part of the device scene a refit rewrites. The master set belongs to the handle (vertex updates, refit, ray
// queries and export work on it). When the scene has dynamic meshes AND several frames are in flight, every frame
// context owns another set, brought up to date (vertex copy + refit) when a frame is submitted on it: a frame that is
// still rendering never sees its tree or vertices change.
struct SceneCopy {
    RpScene dscene;                        // what the kernels get (static arrays are shared between all copies)
    RptrBvh4Node *nodes = nullptr;
    RptrBvhTri *tris = nullptr;
    RpShadeTri *shade = nullptr;            // one shading record per triangle (dshade.h): a refit rewrites the positions of dynamic meshes' records
    float *node_box = nullptr, *tri_box = nullptr, *inst_box = nullptr;
    std::vector<float *> dynpos;            // per global geometry: float positions (9 per triangle) or NULL
    std::vector<const float **> mesh_dyn;   // per mesh: device table of its geometries' dynpos pointers
    std::vector<char> mesh_dirty;           // 0 clean, 1 new vertices, 2 dynamic but triangle bounds never written
    uint64_t version = 0;                   // rptr_hip.refit_version this copy reflects
    // refit of the dynamic bottom-level trees by depth levels (lbvh.h): the node list (every mesh's nodes in its own slice, deepest
    // level first), per mesh RP_REFIT_LEVELS [begin, end) pairs, per mesh the node count
    uint32_t *blas_list = nullptr;
    uint2 *blas_levels = nullptr;
    int *mesh_count = nullptr;
    std::vector<std::array<uint2, RP_REFIT_LEVELS>> host_levels; // per mesh: the level table as the host knows it
    std::vector<char> levels_known;         // per mesh: host_levels is current (a device-built tree: once its read-back has arrived)
    std::vector<uint2 *> pinned_levels;     // per dynamic mesh: pinned staging of that read-back
    std::vector<hipEvent_t> ev_levels;
    std::vector<char> device_built;         // per mesh: its tree was rebuilt on the device (node count lives in mesh_count)
    std::vector<uint64_t> built_epoch;      // per mesh: rptr_hip.rebuild_epoch this copy's tree reflects
    RpLbvhScratch scratch;                  // work space of device-side rebuilds (allocated at the first one)
};

struct Span {
    hipEvent_t a, b;
    int kind; // 0 extend, 1 connect, 2 shade, 3 tail, 4 resolve, 5 other (regrouping pass)
};

// Everything one frame in flight owns: its stream, path state, queues, counters, stack scratch, events.
// frames_in_flight == 1: the single context runs on the backend's stream (rptr_hip.stream) and resolves straight
// into the accumulation buffer. > 1: every context has its own stream; the latency-bound tail of frame i (late
// bounces) overlaps the head of frame i+1, resolves stay ordered, and each context keeps a copy of the image it produced.
struct FrameCtx {
    hipStream_t stream = nullptr;
    bool own_stream = false;
    RpPathState ps = {};
    RpShadowRays sq = {};
    uint32_t *queue[2] = {nullptr, nullptr};
    RpCounters *counters = nullptr;
    RpCounters *host_counters = nullptr; // pinned
    int *gstack = nullptr;
    // the shadow rays of bounce b and the closest-hit rays of bounce b+1 only depend on shade(b): connect runs on a side
    // stream next to the following extend (two latency-bound launches overlap), shade(b+1) waits for both
    hipStream_t side = nullptr;
    int *gstack_side = nullptr;
    hipEvent_t ev_fork = nullptr, ev_side = nullptr;
    float4 *out_accum = nullptr; // frames_in_flight > 1: the image after this frame's resolve (a batch: one image per frame, max_batch_frames of them)
    uchar4 *out_fb = nullptr;
    uint2 *aov[3] = {nullptr, nullptr, nullptr}; // RGBA16F albedo+roughness, normal+depth, motion+jitter of this context's last frame
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_dep = nullptr, ev_resolved = nullptr;
    std::vector<hipEvent_t> ev_pool;
    // the frame in flight on this context
    bool pending = false;        // a frame (or a batch of frames) was submitted here and not all of its tickets have been waited for
    bool synced = false;         // ... and its end has been awaited: stats are in batch_stats
    uint64_t ticket = 0;         // the first ticket of the batch; its frames hold ticket .. ticket + batch_n - 1
    int batch_n = 1;
    uint32_t collected = 0;      // bit k: frame k of the batch has been waited for
    int batch_spp_after[16] = {0};
    RptrStats batch_stats;       // what every frame of the batch reports (totals / batch_n)
    std::vector<Span> spans;
    RpCounters earlier_batches; // counters of the batches that were already synchronised (spp > max_batch_spp)
    int launches_extend = 0, launches_connect = 0, spp_after = 0;
    int tail_from = 0; // the bounce at which this context's last frame handed over to the tail kernel (= max depth: no tail)
    size_t gstack_threads = 0;    // threads the stack scratch is sized for
    // multi-GPU gather (host_comm.h): the image this context produced is being sent; its next frame waits for that on the device
    hipEvent_t ev_gather = nullptr;
    bool gather_pending = false;
};

} // namespace

// ------------------------------------------------------------------ options (include/rptr_hip.h "Options")
// Everything that decides how the library builds and schedules, beyond RptrCreateInfo, is an integer option with a name: set through
// rptr_hip_set_option (h == NULL: the process default new handles start from), read back through rptr_hip_get_option. Each option also
// has an environment variable -- the experimenter's override: when it is set, its value wins over the default AND over rptr_hip_set_option
// (A/B runs of an unmodified host, tools/ab.sh) -- read once per handle, in rptr_hip_create. Nothing else in the library reads the
// environment (GPU_MAX_HW_QUEUES is the HIP runtime's variable, RPTR_FRAMES_IN_FLIGHT overrides RptrCreateInfo.frames_in_flight).
enum RpOpt : int {
    // supported: documented in include/rptr_hip.h "Options", enumerated by rptr_hip_option_count / rptr_hip_option_name
    OPT_FLATTEN, OPT_FLATTEN_MAX_TRIS, OPT_BVH_BUILDER, OPT_DEVICE_BUILD_MIN_TRIS, OPT_TRAVERSE_NODE_MIN, OPT_TRAVERSE_REFILL_MIN, OPT_SINGLE_INSTANCE, OPT_MAX_BATCH_FRAMES,
    OPT_MAX_BATCH_SPP, OPT_PATH_BUDGET_MB, OPT_BLOCKS_PER_CU, OPT_SIDE_CONNECT, OPT_AOVS, OPT_TAIL_BOUNCE, OPT_TAIL_THRESHOLD,
    OPT_STAGE_TIMING, OPT_COMM_TRANSPORT, OPT_COMM_PRIORITY, OPT_QUIET, OPT_TRAVERSE_FETCH, OPT_FAST_MATH,
    OPT_PUBLIC_COUNT,
    // experiments that were measured and not adopted (profiles/r03_notes.md, r05_notes.md): reachable as "experimental.<key>" and through their
    // environment variables, not enumerated, no promise that they stay
    OPT_REBRAID = OPT_PUBLIC_COUNT, OPT_TLAS_COLLAPSE, OPT_COLLAPSE, OPT_PRESPLIT_DENSITY, OPT_PRESPLIT_BUDGET_PCT, OPT_HOST_PLOC, OPT_PLOC_TOP, OPT_PLOC_LEAF, OPT_LDS_TOP, OPT_REGROUP, OPT_COMM_SELF,
    OPT_COUNT
};
struct RpOptDesc {
    const char *key, *env; // env: atoll of the variable unless parse_option_env knows better (names, pairs)
    long long def, lo, hi;
};
static const RpOptDesc g_opt_desc[OPT_COUNT] = {
    {"flatten", "RPTR_FLATTEN", -1, -1, 1},                       // -1 auto: static multi-instance scenes become ONE world-space tree; 0 never; 1 = auto (kept for old hosts)
    {"flatten_max_tris", "RPTR_FLATTEN_MAX_TRIS", 1ll << 26, 0, 1ll << 31}, // ... up to this many instanced triangles (~150 bytes each)
    {"bvh_builder", "RPTR_BVH_BUILDER", 0, 0, 2},                 // 0 auto, 1 host (binned SAH), 2 device (PLOC)
    {"device_build_min_tris", "RPTR_DEVICE_BUILD_MIN_TRIS", 2ll << 20, 0, 1ll << 31},
    {"traverse_node_min", "RPTR_TRAVERSE_PRESET", -1, -1, 64},    // dtraverse.h thresholds; -1: chosen per scene at set_scene
    {"traverse_refill_min", nullptr, -1, -1, 64},
    {"single_instance", "RPTR_NO_SINGLE_INSTANCE", 1, 0, 1},      // queries of scenes with one instance record start inside it
    {"max_batch_frames", "RPTR_MAX_BATCH_FRAMES", 8, 1, 16},      // frames (output images) a launch sequence may hold          [initialize]
    {"max_batch_spp", "RPTR_MAX_BATCH_SPP", 0, 0, 64},            // sample slots in flight per frame context; 0: from the budget [initialize]
    {"path_budget_mb", "RPTR_PATH_BUDGET_MB", 6144, 1, 1 << 20},  // path state per frame context                                [initialize]
    {"blocks_per_cu", "RPTR_BLOCKS_PER_CU", 0, 0, 16},            // persistent traversal blocks per CU; 0: occupancy / contexts [initialize]
    {"side_connect", "RPTR_SIDE_CONNECT", -1, -1, 1},             // connect(b) beside extend(b+1); -1: on for one frame context [initialize]
    {"aovs", "RPTR_AOVS", 1, 0, 1},                               //                                                             [initialize]
    {"tail_bounce", "RPTR_TAIL_BOUNCE", -1, -1, RP_MAX_BOUNCES},  // -1 adaptive, 0 no tail kernel, k: from bounce k
    {"tail_threshold", "RPTR_TAIL_THRESHOLD", 65536, 0, 1 << 30},
    {"stage_timing", "RPTR_STAGE_TIMING", 0, 0, 2},               // events per stage for RptrStats.*_time_ms: a diagnostic (level 2: ~0.06 ms per 1080p frame)
    {"comm_transport", "RPTR_COMM_TRANSPORT", 0, 0, 3},           // 0 auto, 1 rccl, 2 copy, 3 peer                              [comm init]
    {"comm_priority", "RPTR_COMM_PRIORITY", 1, 0, 1},
    {"quiet", "RPTR_QUIET", 0, 0, 1},
    {"traverse_fetch", "RPTR_TRAVERSE_FETCH", 0, 0, 4096},        // queue entries a traversal wave takes per pool at most (multiple of 64); 0: per scene, with the thresholds
    {"fast_math", "RPTR_FAST_MATH", 0, 0, 1},                     // the shading stages' division / square root: 0 IEEE (the oracle's bits), 1 the hardware's 1-ulp rcp / sqrt / rsq (dmath.h)
    // ---- experimental.<key>
    {"rebraid", "RPTR_REBRAID", 0, 0, 64},                        // instance records per instance in the top level; 0 auto (4 from 16 instances on)
    {"tlas_collapse", "RPTR_TLAS_COLLAPSE", 0, 0, 2},             // rptr::COLLAPSE_* of the top level
    {"collapse", "RPTR_COLLAPSE", -1, -1, 2},                     // rptr::COLLAPSE_* of the bottom-level trees; -1: per tree (bvh_build.h)
    {"presplit_density", "RPTR_PRESPLIT", 0, 0, 1 << 30},         // triangle pre-splitting of host-built static trees (0 off)
    {"presplit_budget_pct", nullptr, 100, 0, 10000},              // ... extra references allowed, % of the triangle count
    {"host_ploc", "RPTR_HOST_PLOC", 0, 0, 1024},                  // > 0: the host states the device builder's clustering with this radius
    {"ploc_top", "RPTR_PLOC_TOP", 0, 0, 1ll << 31},               // clusters at which the PLOC clustering stops (0: RP_PLOC_TOP)
    {"ploc_leaf", "RPTR_PLOC_LEAF", 0, 0, 7},
    {"lds_top", "RPTR_LDS_TOP", 0, 0, 1},
    {"regroup_materials", "RPTR_REGROUP", 0, 0, 1},
    {"comm_self", "RPTR_COMM_SELF", 0, 0, 1},
};
struct RpOptions {
    long long v[OPT_COUNT];
    bool from_env[OPT_COUNT];
};
// the process defaults (rptr_hip_set_option(NULL, ..)): hosts with one thread per GPU create handles side by side, so reads and writes go
// through one lock and readers get a copy
static std::mutex &process_default_lock() {
    static std::mutex m;
    return m;
}
static RpOptions &process_default_storage() {
    static RpOptions o = [] {
        RpOptions d;
        for (int k = 0; k < OPT_COUNT; ++k) {
            d.v[k] = g_opt_desc[k].def;
            d.from_env[k] = false;
        }
        return d;
    }();
    return o;
}
static RpOptions process_default_options() {
    std::lock_guard<std::mutex> g(process_default_lock());
    return process_default_storage();
}
static void set_process_default_option(int k, long long value) {
    std::lock_guard<std::mutex> g(process_default_lock());
    process_default_storage().v[k] = value;
}
static int find_option(const char *key) {
    if (!key) return -1;
    const bool experimental = !strncmp(key, "experimental.", 13);
    if (experimental) key += 13;
    for (int k = experimental ? (int)OPT_PUBLIC_COUNT : 0; k < (experimental ? (int)OPT_COUNT : (int)OPT_PUBLIC_COUNT); ++k)
        if (!strcmp(key, g_opt_desc[k].key)) return k;
    return -1;
}
static long long clamp_option(int k, long long v) { return std::max(g_opt_desc[k].lo, std::min(g_opt_desc[k].hi, v)); }
// the environment's word on every option (names and pairs where the variable always took them)
static void apply_option_env(RpOptions &o) {
    auto set = [&](int k, long long v) {
        o.v[k] = clamp_option(k, v);
        o.from_env[k] = true;
    };
    auto collapse_rule = [](const char *e) -> long long {
        if (!strcmp(e, "even")) return 1;
        if (!strcmp(e, "dp") || !strcmp(e, "optimal")) return 2;
        if (!strcmp(e, "greedy")) return 0;
        return atoll(e);
    };
    for (int k = 0; k < OPT_COUNT; ++k) {
        const char *e = g_opt_desc[k].env ? getenv(g_opt_desc[k].env) : nullptr;
        if (!e) continue;
        switch (k) {
        case OPT_BVH_BUILDER: set(k, !strcmp(e, "host") ? 1 : !strcmp(e, "device") ? 2 : !strcmp(e, "auto") ? 0 : atoll(e)); break;
        case OPT_TLAS_COLLAPSE: set(k, collapse_rule(e)); break;
        case OPT_COLLAPSE: set(k, !strcmp(e, "") ? -1 : collapse_rule(e)); break;
        case OPT_PRESPLIT_DENSITY: // "density[,budget]"
            set(k, (long long)atof(e));
            if (const char *c = strchr(e, ',')) set(OPT_PRESPLIT_BUDGET_PCT, (long long)(atof(c + 1) * 100.0 + 0.5));
            break;
        case OPT_TRAVERSE_NODE_MIN: // "node_min,refill_min"
            set(k, atoll(e));
            if (const char *c = strchr(e, ',')) set(OPT_TRAVERSE_REFILL_MIN, atoll(c + 1));
            else set(OPT_TRAVERSE_REFILL_MIN, 0);
            break;
        case OPT_SINGLE_INSTANCE: set(k, 0); break; // RPTR_NO_SINGLE_INSTANCE: its presence switches the shortcut off
        case OPT_COMM_TRANSPORT: set(k, !strcmp(e, "rccl") ? 1 : !strcmp(e, "copy") ? 2 : !strcmp(e, "peer") ? 3 : atoll(e)); break;
        default: set(k, atoll(e)); break;
        }
    }
}
// what a handle-less entry point (rptr_hip_build_bvh_host) works with: the process defaults under the environment
static RpOptions effective_default_options() {
    RpOptions o = process_default_options();
    apply_option_env(o);
    return o;
}


struct RptrComm; // host_comm.h

struct rptr_hip {
    RpOptions opt; // rptr_hip_set_option / the environment's overrides (rptr_hip_create)
    RptrComm *comm = nullptr; // communicator rank of this handle (rptr_hip_comm_init_rank / _init_all), NULL on a single GPU
    std::string last_error;
    int device = 0;
    int rank = 0, world = 1, stripe_rows = 32;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int num_cus = 0;
    size_t bytes_allocated = 0, bytes_frame = 0, bytes_scene = 0; // what is allocated now (frame buffers + path state, scene)
    double bvh_build_ms = 0.0, bvh_device_ms = 0.0; // the last set_scene: its acceleration-structure step, the device part of it
    double bvh_area_cost = 0.0;                     // surface-area cost of the largest bottom-level tree (picks the traversal's scheduling thresholds)
    bool bvh_device_built = false;
    std::vector<void *> allocations;

    // frame
    int width = 0, height = 0, local_rows = 0;
    int tiles_x = 0, tiles_y = 0, npix_padded = 0;
    int max_batch_spp = 0;
    uint32_t frame_id = 0, frame_offset = 0;
    int accumulated_spp = 0;

    RptrRenderParams params;
    RptrSceneParams scene_params;
    RptrLightSamplingConfig lighting;
    bool have_params = false;

    // scene
    bool have_scene = false;
    std::vector<RptrBvh4Node> h_nodes;
    std::vector<std::array<float, 6>> h_node_box; // exact float bounds of every node
    int num_tlas_nodes = 0;
    std::vector<RptrBvhTri> h_tris;
    std::vector<RptrBvhInstance> h_insts;
    int num_tlas_insts = 0;
    std::vector<MeshRt> meshes;
    std::vector<void *> scene_allocs;
    int num_lights = 0, num_materials = 0;
    bool uses_textures = false;
    bool uses_alpha = false; // some material lacks BASE_MATERIAL_NOALPHA: extend/connect run the any-hit alpha test // some material has a textured parameter or a normal map
    // dynamic meshes (Mesh::Dynamic: float vertex buffer + BLAS update + TLAS refit, render_vulkan.cpp:942-952,1323-1354)
    SceneCopy master;                       // dscene + the refit targets of the handle
    std::vector<SceneCopy> ctx_scene;       // one per frame context when the scene is dynamic and frames_in_flight > 1
    uint64_t refit_version = 0;             // bumped by every rptr_hip_refit that changed something
    std::vector<uint32_t> geom_tris;        // per global geometry: triangle count
    std::vector<int> geom_mesh;             // per global geometry: owning mesh
    std::vector<int> mesh_root;             // per mesh: absolute node index of the BLAS root
    uint32_t *d_refit_list = nullptr;       // top-level node indices (bit 31 set) by height
    std::vector<std::array<uint32_t, 2>> refit_levels_tlas; // [begin, end) per height
    uint2 *d_refit_levels = nullptr;        // the same pairs on the device
    bool refit_top_all = false;             // instance bounds + top-level levels fit one single-block launch (rp_k_refit_top)
    bool has_dynamic = false;               // some mesh is dynamic
    std::vector<int> mesh_geometry_base;    // per mesh: first geometry record of the FIRST parameterized mesh that uses it (-1: none does)
    size_t flat_tris = 0, flat_nodes = 0;   // the scene's static instances were built as one world-space tree (option "flatten"): its triangles / nodes come first
    // BVH policy (RenderBackendOptions::force_bvh_rebuild / rebuild_triangle_budget, librender/render_params.glsl.h:61,90-93)
    bool bvh_force_rebuild = false;
    long long bvh_budget = 0, bvh_credit = 0; // triangles a refit call may rebuild; what has been saved up
    std::vector<uint64_t> rebuild_epoch;    // per mesh: bumped when the policy asks for a rebuild of its tree
    int rebuild_cursor = 0;                 // round robin over the dynamic meshes
    uint64_t rebuilds_done = 0;
    uint64_t rebuild_failures = 0;          // device-side rebuilds that could not start (their meshes were refitted instead)
    bool host_bvh_stale = false;
    uint64_t vertex_updates = 0, vertex_updates_refitted = 0;
    bool master_refit_pending = false; // rptr_hip_refit with frame contexts that own their sets: the master tree is refitted on demand

    // device buffers (frame sized)
    std::vector<FrameCtx> ctx;      // frames in flight (RptrCreateInfo.frames_in_flight, at least 1)
    uint64_t next_ticket = 1;
    int next_ctx = 0;
    int output_ctx = -1;            // frames_in_flight > 1: the context whose image read-backs return (last waited frame)
    int output_index = 0;           // ... and which frame of that context's batch
    int max_batch_frames = 8;       // option "max_batch_frames": per-frame output images a context keeps (rptr_hip_render_batch_async)
    int aov_ctx = 0;                // the context whose AOV images readback_aov returns (last finished frame)
    bool output_overwritten = false; // a newer frame was submitted on output_ctx / aov_ctx: its resolve rewrites the images a read-back
    bool aov_overwritten = false;    // would return, so read-backs fail until that frame has been waited for
    int tail_mode = -1;             // RPTR_TAIL_BOUNCE: -1 adaptive, 0 off, k > 0: the tail kernel takes over at bounce k
    int tail_adaptive = 1 << 30;    // adaptive choice for the next frame (from the queue lengths of the last finished frame)
    int tail_blocks = 0;
    int tail_threshold = 65536;     // RPTR_TAIL_THRESHOLD: queue length below which a bounce goes to the tail kernel
    // ray queries on device buffers (enable_ray_queries / render_ray_queries: the reference's ray_query_buffer / ray_result_buffer)
    RptrRenderRayQuery *rq_queries = nullptr;
    float4 *rq_results = nullptr;
    size_t rq_capacity = 0;
    bool lights_disabled = false;   // light_sampling_variant == LIGHT_SAMPLING_VARIANT_NONE: no area-light NEE (rptr_hip_set_light_sampling_variant)
    bool aovs = true;               // the reference writes its AOV images with every frame (ENABLE_AOV_BUFFERS, render_vulkan.cpp:2083-2086)
    RptrCamera prev_camera;         // the previous frame's view (VP_reference)
    bool have_prev_camera = false;
    hipEvent_t last_resolved = nullptr; // resolve of the most recently submitted frame (resolves run in submission order)
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {1, 1, 1};
    float4 *accum = nullptr;
    uchar4 *fb = nullptr;
    size_t path_capacity = 0;
    int persistent_blocks = 0;
    int extend_later_blocks = 0;     // grid of a closest-hit launch of bounce >= 1 (RP_EXTEND_LATER_WAVES)
    int alone_blocks[4] = {0, 0, 0, 0}; // two frame contexts: the grids (first / later closest-hit, shadow rays [two-level, one record]) of a frame that is alone on the GPU
    int connect_blocks[2] = {0, 0};  // grid of a stand-alone shadow-ray launch, [single instance record ? 1 : 0] (RP_CONNECT_WAVES)

    // options (environment, read once)
    bool side_only_alone = false; // side_connect chosen by the library for a handle with two frame contexts: only for a frame that is alone on the GPU
    int side_connect = 0; // connect(b) on a side stream next to extend(b+1): the default for handles with ONE frame context (RPTR_SIDE_CONNECT=0|1 overrides)
    int stage_timing = 2; // hipEvent pairs per frame: 0 none, 1 around the closest-hit traversal launches, 2 every stage
    bool freeze_frame = false; // RenderConfiguration::freeze_frame: frame_offset / frame_id stand still
    int rng_variant = RPTR_RNG_VARIANT_UNIFORM; // rptr_hip_set_rng_variant
    uint32_t *rng_table = nullptr;              // device copy of SobolData / BNData (hipMalloc, freed on replace / destroy)

    RptrStats stats;
};

namespace {

void comm_release(rptr_hip *h); // host_comm.h

int fail(rptr_hip *h, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->last_error = buf;
    g_last_error = buf;
    return code;
}

// the members the frame loop reads per frame follow the options (the rest is read where it takes effect: initialize, set_scene, comm init)
void sync_options(rptr_hip *h) {
    h->tail_mode = (int)h->opt.v[OPT_TAIL_BOUNCE];
    h->tail_threshold = (int)h->opt.v[OPT_TAIL_THRESHOLD];
    h->stage_timing = (int)h->opt.v[OPT_STAGE_TIMING];
}

// Hardware queues. Every frame context renders on a stream of its own, and the HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware
// queues (default 4): streams that share a queue serialise, and the schedule bench.py measures (11 contexts) needs one queue per
// context + the caller's stream + the communication stream. The runtime reads the variable ONCE, when the process makes its first HIP
// call. The variable belongs to the HOST: the library edits it only when the host says so (RptrCreateInfo.flags &
// RPTR_CREATE_SET_HW_QUEUES: bin/rptr_hip does; round 5 did it from a load-time constructor, a surprise for an embedding host), and then
// only in a create that may still be the process's first HIP call. Otherwise it reads the variable and says once on stderr when the frame
// contexts outnumber the queues (option "quiet" silences it).
static bool g_hw_queues_set_by_library = false;
// did the host initialise HIP before this library could set the variable? hipGetDeviceCount-style calls do not tell; what does: whether a
// primary context is already active on device 0 when the first handle is created
static bool hip_was_initialised_before_us() {
    unsigned int flags = 0;
    int active = 0;
    return hipDevicePrimaryCtxGetState(0, &flags, &active) == hipSuccess && active != 0;
}

// may_set: RptrCreateInfo.flags & RPTR_CREATE_SET_HW_QUEUES -- the host lets this create edit the process's environment. Without it the
// library only reads the variable and says (once, on stderr) when the contexts outnumber the queues.
static void ensure_hw_queues(int frames_in_flight, bool may_set) {
    static bool first_create = true;
    if (const char *s = getenv("RPTR_FRAMES_IN_FLIGHT")) frames_in_flight = atoi(s);
    const int want = std::max(1, std::min(frames_in_flight, 16)) + 2; // + the caller's stream + the communication stream
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    const int have = e ? atoi(e) : 4;
    if (have < want) {
        // ours to raise when the host said so and the variable is unset (or was set by an earlier create of this library). The setenv comes
        // BEFORE any HIP call of this create: the first one makes the runtime read the variable.
        if (may_set && first_create && (!e || g_hw_queues_set_by_library)) {
            char buf[16];
            snprintf(buf, sizeof buf, "%d", std::max(want, 16));
            setenv("GPU_MAX_HW_QUEUES", buf, 1);
            g_hw_queues_set_by_library = true;
            if (hip_was_initialised_before_us() && effective_default_options().v[OPT_QUIET] == 0)
                fprintf(stderr, "rptr_hip: RPTR_CREATE_SET_HW_QUEUES came too late -- the process already uses HIP with GPU_MAX_HW_QUEUES=%d; %d frame contexts "
                                "want %d hardware queues (streams that share a queue serialise)\n", have, want - 2, want);
        } else if (frames_in_flight > 1 && effective_default_options().v[OPT_QUIET] == 0) {
            static bool warned = false;
            if (!warned)
                fprintf(stderr, "rptr_hip: GPU_MAX_HW_QUEUES=%d but %d frame contexts want %d hardware queues (streams that share a queue serialise); "
                                "set GPU_MAX_HW_QUEUES>=%d before the process's first HIP call%s\n", have, want - 2, want, want,
                        may_set ? "" : ", or pass RPTR_CREATE_SET_HW_QUEUES in RptrCreateInfo.flags from a process that has not used HIP yet");
            warned = true;
        }
    }
    first_create = false;
}

#define HIP_TRY(h, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return fail(h, RPTR_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

template <class T>
int dev_alloc(rptr_hip *h, T **out, size_t count, std::vector<void *> *track) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(h, RPTR_E_NOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    (track == &h->scene_allocs ? h->bytes_scene : h->bytes_frame) += bytes;
    h->bytes_allocated = h->bytes_scene + h->bytes_frame;
    (track ? track : &h->allocations)->push_back(p);
    *out = reinterpret_cast<T *>(p);
    return RPTR_OK;
}

void release_scene_copy_host(SceneCopy &sc) { // pinned staging + events of the level read-backs
    for (uint2 *p : sc.pinned_levels)
        if (p) (void)hipHostFree(p);
    for (hipEvent_t e : sc.ev_levels)
        if (e) (void)hipEventDestroy(e);
    sc.pinned_levels.clear();
    sc.ev_levels.clear();
}

void free_list(std::vector<void *> &v) {
    for (void *p : v) (void)hipFree(p);
    v.clear();
}

// rows owned by `rank`: stripes s with s % world == rank
int local_row_count(int height, int stripe_rows, int rank, int world) {
    int n_stripes = (height + stripe_rows - 1) / stripe_rows;
    int rows = 0;
    for (int s = rank; s < n_stripes; s += world) rows += std::min(stripe_rows, height - s * stripe_rows);
    return rows;
}

// inverse of a row-major 3x4 affine transform; cofactors in double, rounded once
void invert_affine(const float m[12], float out[12]) {
    double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], hh = m[9], i = m[10];
    double A = e * i - f * hh, B = -(d * i - f * g), C = d * hh - e * g;
    double det = a * A + b * B + c * C;
    double id = 1.0 / det;
    double r[9] = {A * id, -(b * i - c * hh) * id, (b * f - c * e) * id, B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                   C * id, -(a * hh - b * g) * id, (a * e - b * d) * id};
    double tx = m[3], ty = m[7], tz = m[11];
    for (int k = 0; k < 3; ++k) {
        out[4 * k + 0] = (float)r[3 * k + 0];
        out[4 * k + 1] = (float)r[3 * k + 1];
        out[4 * k + 2] = (float)r[3 * k + 2];
        out[4 * k + 3] = (float)(-(r[3 * k + 0] * tx + r[3 * k + 1] * ty + r[3 * k + 2] * tz));
    }
}

// librender/dequantize.glsl:8-21 on the host: the BLAS is built from dequantised
// floats exactly as the reference feeds them to the driver (render_vulkan.cpp:698-711)
inline void dequantize_position(uint64_t w, const float sc[3], const float of[3], float out[3]) {
    out[0] = float(uint32_t(w) & 0x1FFFFFu) * sc[0] + of[0];
    out[1] = float(uint32_t(w >> 21) & 0x1FFFFFu) * sc[1] + of[1];
    out[2] = float(uint32_t(w >> 42) & 0x1FFFFFu) * sc[2] + of[2];
}

hipEvent_t next_event(FrameCtx &c, size_t &cursor) {
    if (cursor >= c.ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c.ev_pool.push_back(e);
    }
    return c.ev_pool[cursor++];
}

int grid_for(const rptr_hip *h, size_t n, int per_cu = 8) {
    size_t blocks = (n + 255) / 256;
    size_t cap = (size_t)h->num_cus * per_cu;
    return (int)std::max<size_t>(1, std::min(blocks, cap));
}

} // namespace

extern "C++" {
// ------------------------------------------------------------------ host side of the acceleration structure
// Everything of set_scene that needs no device: per-mesh binned-SAH trees from the dequantised triangles, the
// top level over the instance bounds, the 4-wide collapse, the 64-byte encoding, and the worst-case stack need.
// Also reachable without a GPU through rptr_hip_build_bvh_host (CPU tests walk this tree with the oracle).
struct HostBvh {
    std::vector<RptrBvh4Node> nodes;
    std::vector<std::array<float, 6>> node_box;
    std::vector<RptrBvhTri> tris;
    std::vector<RptrBvhInstance> insts;
    std::vector<MeshRt> meshes;
    std::vector<int> mesh_root;
    int num_tlas_nodes = 0;
    int num_tlas_insts = 0; // instance records the top level refers to (a flattened scene keeps the scene's own records behind them)
    float scene_lo[3] = {0, 0, 0}, scene_hi[3] = {1, 1, 1};
    int stack_need = 0;
    size_t flat_tris = 0, flat_nodes = 0; // a (partially) flattened scene: triangles / nodes of its one world-space tree (they come first)
    int flat_id_bias = 0;                 // ... and where its triangles' own instance records start (record = bias + instance id)
    bool device_built = false; // some bottom-level tree came from the device builder (ploc.h)
    double device_ms = 0.0;
    int device_iterations = 0;
};

// Flattening (option "flatten", default auto): a static scene with several instances is built as ONE bottom-level tree over all instanced triangles,
// pre-transformed to world space (a 10 M-triangle forest is 0.6 GB of triangles and nodes: nothing on a 288 GB device). Rays then
// meet one well-separated tree instead of a thousand overlapping instance boxes, each with its own ray transform. Hits are
// found on the world-space triangles, so t / u / v may differ from the two-level walk by rounding; shading still reads the
// mesh's own vertex streams through the instance record the triangle names (RptrBvhTri.flags bits 8..31).
// What both set_scene and rptr_hip_build_bvh_host check before they touch the borrowed arrays: index ranges of the mesh /
// geometry / material tables (a malformed .vks file must be rejected, not read out of bounds). Returns "" when fine.
static std::string validate_scene_tables(const RptrSceneDesc *s) {
    char buf[256];
    auto err = [&](const char *fmt, auto... a) {
        snprintf(buf, sizeof(buf), fmt, a...);
        return std::string(buf);
    };
    if ((s->num_geometries && !s->geometries) || (s->num_meshes && !s->meshes) || (s->num_parameterized_meshes && !s->parameterized_meshes) ||
        (s->num_instances && !s->instances) || (s->num_materials && !s->materials) || (s->num_lights && !s->lights))
        return "a table of the scene is NULL but its count is not 0";
    for (uint32_t g = 0; g < s->num_geometries; ++g)
        if (s->geometries[g].num_tris && !s->geometries[g].qpos) return err("geometry %u: qpos is NULL", g);
    for (uint32_t m = 0; m < s->num_meshes; ++m)
        if ((uint64_t)s->meshes[m].first_geometry + s->meshes[m].num_geometries > s->num_geometries)
            return err("mesh %u: geometries [%u, +%u) are outside the scene's %u geometries", m, s->meshes[m].first_geometry, s->meshes[m].num_geometries,
                       s->num_geometries);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        if (pm.mesh >= s->num_meshes) return err("parameterized mesh %u: bad mesh index", p);
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        if (mesh.num_geometries && !pm.material_offsets) return err("parameterized mesh %u: material_offsets is NULL", p);
        size_t off = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t nt = s->geometries[mesh.first_geometry + j].num_tris;
            if (pm.material_offsets[j] < 0 || (uint32_t)pm.material_offsets[j] >= s->num_materials)
                return err("parameterized mesh %u geometry %u: material offset %d out of range (%u materials)", p, j, pm.material_offsets[j], s->num_materials);
            if (pm.tri_material_ids) {
                uint32_t max_local = 0;
                for (uint32_t t = 0; t < nt; ++t) max_local = std::max<uint32_t>(max_local, pm.tri_material_ids[off + t]);
                if (nt && (uint64_t)pm.material_offsets[j] + max_local >= s->num_materials)
                    return err("parameterized mesh %u geometry %u: per-triangle material id %u + offset %d is outside the scene's %u materials", p, j, max_local,
                               pm.material_offsets[j], s->num_materials);
            }
            off += nt;
        }
    }
    for (uint32_t i = 0; i < s->num_instances; ++i)
        if (s->instances[i].parameterized_mesh >= s->num_parameterized_meshes) return err("instance %u: bad mesh", i);
    return std::string();
}

// 0: two-level; 1: the whole scene is one world-space tree (every instanced mesh is static); 2: PARTIAL -- the scene has dynamic meshes: the
// instances of its static meshes are flattened into one tree, which the top level holds as one identity instance beside the records of the
// dynamic meshes' instances (round 5: a forest with one animated character used to fall back to the two-level walk as a whole: 1.5 x)
static int want_flatten(const RptrSceneDesc *s, const RpOptions &o) {
    // option "flatten": -1 / 1 = every static multi-instance scene that fits "flatten_max_tris" (the default: the library knows which
    // meshes are dynamic -- RptrMeshDesc.dynamic, the reference's per-mesh build intent, vulkan/render_vulkan.cpp:942-952 -- and a flattened
    // tree is 1.5-1.6 x faster to trace than the two-level one); 0 = never
    if (o.v[OPT_FLATTEN] == 0 || s->num_instances < 2) return 0;
    const size_t limit = (size_t)o.v[OPT_FLATTEN_MAX_TRIS];
    size_t total = 0;
    uint32_t n_static = 0, n_dynamic = 0;
    for (uint32_t i = 0; i < s->num_instances; ++i) {
        const RptrMeshDesc &mesh = s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh].mesh];
        if (mesh.dynamic) {
            ++n_dynamic;
            continue;
        }
        ++n_static;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) total += s->geometries[mesh.first_geometry + j].num_tris;
    }
    if (total > limit || total == 0 || (uint64_t)s->num_instances + n_dynamic + 2 >= (1u << 24)) return 0;
    if (n_dynamic == 0) return 1;
    return n_static >= 2 ? 2 : 0;
}

// ------------------------------------------------------------------ device-side build of one bottom-level tree (ploc.h)
// What a build hands back to build_host_bvh: the tree in the form the host builder's encode_tree produces (local node indices from 0,
// leaf ranges from triangle 0 of `tris`), so that everything behind it -- top level, relocation, stack need, upload -- is shared.
struct DeviceTree {
    std::vector<RptrBvh4Node> nodes;
    std::vector<std::array<float, 6>> boxes;
    std::vector<RptrBvhTri> tris;
    double ms_device = 0.0, ms_top = 0.0;
    int iterations = 0;
    uint32_t top_clusters = 0;
};
// segments: the triangles' sources (device pointers of the vertex streams the scene upload made); mat_alpha: per material, 1 = alpha-tested
using DeviceTreeBuilder = std::function<bool(const std::vector<RpBuildSegment> &, uint32_t, DeviceTree &)>;
struct DeviceBuildCtx {
    DeviceTreeBuilder build;                         // empty: no device (rptr_hip_build_bvh_host on a CPU box)
    const std::vector<const uint64_t *> *d_qpos = nullptr; // per global geometry
    const std::vector<RpGeomRecord> *geoms = nullptr; // per (parameterized mesh, geometry): mat_ids
    size_t min_tris = (size_t)2 << 20;               // RPTR_BVH_BUILDER=auto: prim sets of at least this size are built on the device
};

namespace {
struct DevScratch { // frees what it allocated when the build is over
    std::vector<void *> ptrs;
    template <class T>
    T *get(size_t count) {
        void *p = nullptr;
        if (hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return nullptr;
        ptrs.push_back(p);
        return (T *)p;
    }
    ~DevScratch() {
        for (void *p : ptrs) (void)hipFree(p);
    }
};
} // namespace

static bool device_build_tree(rptr_hip *h, const std::vector<RpBuildSegment> &segs, uint32_t n, const std::vector<uint8_t> &mat_alpha, DeviceTree &out) {
    if (n < 2 || segs.empty()) return false;
    hipStream_t st = h->stream;
    DevScratch S;
#define DB_TRY(expr)                                                                                  \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess) {                                                                       \
            fail(h, RPTR_E_HIP, "device BVH build: %s failed: %s", #expr, hipGetErrorString(_e));    \
            (void)hipGetLastError();                                                                  \
            return false;                                                                             \
        }                                                                                             \
    } while (0)
#define DB_ALLOC(var, T, count)                                                          \
    T *var = S.get<T>(count);                                                            \
    if (!var) {                                                                          \
        fail(h, RPTR_E_NOMEM, "device BVH build: out of device memory (%s)", #var);     \
        (void)hipGetLastError();                                                         \
        return false;                                                                    \
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    struct EvGuard {
        hipEvent_t &a, &b;
        ~EvGuard() {
            if (a) (void)hipEventDestroy(a);
            if (b) (void)hipEventDestroy(b);
        }
    } ev_guard{e0, e1};
    (void)hipEventRecord(e0, st);
    const int g = grid_for(h, n);
    // 1. triangles + bounds
    DB_ALLOC(d_segs, RpBuildSegment, segs.size());
    DB_ALLOC(d_alpha, uint8_t, mat_alpha.size());
    DB_ALLOC(tris_a, RptrBvhTri, (size_t)n + 2);
    DB_ALLOC(tris_b, RptrBvhTri, (size_t)n + 2);
    DB_ALLOC(box_a, float, 6 * (size_t)n);
    DB_ALLOC(box_b, float, 6 * (size_t)n);
    DB_TRY(hipMemcpyAsync(d_segs, segs.data(), segs.size() * sizeof(RpBuildSegment), hipMemcpyHostToDevice, st));
    if (!mat_alpha.empty()) DB_TRY(hipMemcpyAsync(d_alpha, mat_alpha.data(), mat_alpha.size(), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rp_k_build_tris, dim3(g), dim3(256), 0, st, d_segs, (int)segs.size(), n, d_alpha, (uint32_t)mat_alpha.size(), tris_a, box_a);
    // 2. Morton order
    DB_ALLOC(keys_a, unsigned long long, n);
    DB_ALLOC(keys_b, unsigned long long, n);
    DB_ALLOC(bounds, uint32_t, 8);
    size_t sort_bytes = 0, scan_bytes = 0, scan64_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_a, keys_b, (int)n, 0, 64, st);
    DB_ALLOC(flag, uint32_t, n);
    DB_ALLOC(slot, uint32_t, n);
    DB_ALLOC(packed, unsigned long long, n);
    DB_ALLOC(pscan, unsigned long long, n);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, flag, slot, (int)n, st);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan64_bytes, packed, pscan, (int)n, st);
    const size_t cub_bytes = std::max(sort_bytes, std::max(scan_bytes, scan64_bytes)) + 256;
    DB_ALLOC(cub_tmp, char, cub_bytes);
    hipLaunchKernelGGL(rp_k_lbvh_reset, dim3(1), dim3(64), 0, st, bounds);
    hipLaunchKernelGGL(rp_k_lbvh_bounds, dim3(g), dim3(256), 0, st, box_a, n, bounds);
    int index_bits = 1;
    while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
    hipLaunchKernelGGL(rp_k_lbvh_keys, dim3(g), dim3(256), 0, st, box_a, n, bounds, keys_a, index_bits);
    size_t bytes = cub_bytes;
    DB_TRY(hipcub::DeviceRadixSort::SortKeys(cub_tmp, bytes, keys_a, keys_b, (int)n, 0, 64, st));
    hipLaunchKernelGGL(rp_k_lbvh_gather, dim3(g), dim3(256), 0, st, keys_b, n, tris_a, box_a, tris_b, box_b, (1ull << index_bits) - 1ull);
    // 3. PLOC
    DB_ALLOC(cid_a, uint32_t, n);
    DB_ALLOC(cid_b, uint32_t, n);
    DB_ALLOC(nn, uint32_t, n);
    DB_ALLOC(left, int, n);
    DB_ALLOC(right, int, n);
    DB_ALLOC(parent, int, 2 * (size_t)n);
    DB_ALLOC(count, uint32_t, 2 * (size_t)n);
    DB_ALLOC(area, float, n);
    DB_ALLOC(totals, uint32_t, 4);
    DB_ALLOC(dpc, float4, n); // per inner node: the costs the collapse decides by (ploc.h rp_ploc_dp_node)
    float *cbox_a = box_a, *cbox_b = nullptr; // (box_a is free again after the gather; the second cluster box list is its own)
    DB_ALLOC(cbox_second, float, 6 * (size_t)n);
    cbox_b = cbox_second;
    hipLaunchKernelGGL(rp_k_ploc_init, dim3(g), dim3(256), 0, st, n, box_b, cid_a, cbox_a, parent, count);
    uint32_t host_totals[2] = {n, n}; // clusters, nodes made so far (ids below n are the triangles)
    DB_TRY(hipMemcpyAsync(totals, host_totals, sizeof(host_totals), hipMemcpyHostToDevice, st));
    size_t top_k = RP_PLOC_TOP;
    if (h->opt.v[OPT_PLOC_TOP] > 0) top_k = (size_t)h->opt.v[OPT_PLOC_TOP];
    uint32_t m = n, nodes_before = n;
    int iterations = 0;
    while (m > top_k && m > 1) {
        hipLaunchKernelGGL(rp_k_ploc_nn<RP_PLOC_RADIUS>, dim3((m + 255) / 256), dim3(256), 0, st, m, cbox_a, nn);
        hipLaunchKernelGGL(rp_k_ploc_flags, dim3(grid_for(h, m)), dim3(256), 0, st, m, nn, packed);
        bytes = cub_bytes;
        DB_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes, packed, pscan, (int)m, st));
        hipLaunchKernelGGL(rp_k_ploc_apply, dim3(grid_for(h, m)), dim3(256), 0, st, m, n, nn, packed, pscan, cid_a, cbox_a, cid_b, cbox_b, left, right, parent, count, area, totals,
                           totals + 2);
        DB_TRY(hipMemcpyAsync(totals, totals + 2, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
        DB_TRY(hipMemcpyAsync(host_totals, totals, sizeof(host_totals), hipMemcpyDeviceToHost, st));
        DB_TRY(hipStreamSynchronize(st));
        if (host_totals[0] >= m) { // (cannot happen: the globally closest pair is always mutual)
            fail(h, RPTR_E_HIP, "device BVH build: clustering made no progress at %u clusters", m);
            return false;
        }
        if (host_totals[1] > nodes_before) // the merges of this iteration: their children are older, their costs known
            hipLaunchKernelGGL(rp_k_ploc_dp_range, dim3(grid_for(h, host_totals[1] - nodes_before)), dim3(256), 0, st, nodes_before, host_totals[1], n, left, right, count, area, dpc);
        nodes_before = host_totals[1];
        m = host_totals[0];
        std::swap(cid_a, cid_b);
        std::swap(cbox_a, cbox_b);
        ++iterations;
    }
    out.iterations = iterations;
    out.top_clusters = m;
    // 4. the top: binned SAH over the remaining clusters (host, milliseconds), stitched on
    const auto t_top0 = std::chrono::steady_clock::now();
    if (m > 1) {
        std::vector<rptr::BuildPrim> cp(m);
        std::vector<uint32_t> ids(m), cnt(m);
        static_assert(sizeof(rptr::BuildPrim) == 24, "cluster boxes are copied as build primitives");
        DB_ALLOC(d_cnt, uint32_t, m);
        hipLaunchKernelGGL(rp_k_ploc_gather_counts, dim3(grid_for(h, m)), dim3(256), 0, st, m, cid_a, count, d_cnt);
        DB_TRY(hipMemcpyAsync(cp.data(), cbox_a, (size_t)m * 24, hipMemcpyDeviceToHost, st));
        DB_TRY(hipMemcpyAsync(ids.data(), cid_a, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        DB_TRY(hipMemcpyAsync(cnt.data(), d_cnt, (size_t)m * 4, hipMemcpyDeviceToHost, st));
        DB_TRY(hipStreamSynchronize(st));
        rptr::BuiltTree top;
        rptr::build_bvh2(cp.data(), m, 1, 56, 0, top);
        const size_t T = top.nodes.size();
        if (T != (size_t)m - 1) {
            fail(h, RPTR_E_HIP, "device BVH build: the top tree over %u clusters has %zu nodes", m, T);
            return false;
        }
        std::vector<float4> ccost(m), tcost(T); // collapse costs of the cluster roots (from the device) and of the top nodes (made here)
        {
            DB_ALLOC(d_ccost, float4, m);
            hipLaunchKernelGGL(rp_k_ploc_gather_costs, dim3(grid_for(h, m)), dim3(256), 0, st, m, n, cid_a, dpc, d_ccost);
            DB_TRY(hipMemcpyAsync(ccost.data(), d_ccost, (size_t)m * sizeof(float4), hipMemcpyDeviceToHost, st));
            DB_TRY(hipStreamSynchronize(st));
        }
        std::vector<int> tl(T), tr(T);
        std::vector<uint32_t> tc(T);
        std::vector<float> ta(T);
        std::vector<float4> cost_of(T); // by top-tree node index
        std::vector<uint32_t> id_of(T), cnt_of(T);
        const uint32_t first_id = host_totals[1];
        for (int64_t i = (int64_t)T - 1; i >= 0; --i) { // children lie behind their parents: backwards = bottom-up, the root is made last
            const RptrBvhNode &t = top.nodes[(size_t)i];
            const size_t k = T - 1 - (size_t)i;
            uint32_t c_id[2], c_cnt[2];
            const int32_t two[2] = {t.child0, t.child1};
            for (int w = 0; w < 2; ++w) {
                if (two[w] >= 0) {
                    c_id[w] = id_of[(size_t)two[w]];
                    c_cnt[w] = cnt_of[(size_t)two[w]];
                } else {
                    const uint32_t ci = top.order[(size_t)RPTR_BVH_LEAF_FIRST(two[w])];
                    c_id[w] = ids[ci];
                    c_cnt[w] = cnt[ci];
                }
            }
            tl[k] = (int)c_id[0];
            tr[k] = (int)c_id[1];
            tc[k] = c_cnt[0] + c_cnt[1];
            {   // surface (half) area of the node's box = union of its children's boxes, as the clustering computes it for its own nodes
                float lo[3], hi[3];
                for (int a = 0; a < 3; ++a) {
                    lo[a] = std::fmin(t.lo0[a], t.lo1[a]);
                    hi[a] = std::fmax(t.hi0[a], t.hi1[a]);
                }
                const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
                ta[k] = (dx >= 0.0f && dy >= 0.0f && dz >= 0.0f) ? dx * dy + dy * dz + dz * dx : 0.0f;
            }
            {   // ploc.h rp_ploc_dp_node, for a node of the top (same operations in the same order)
                float4 cc[2];
                bool exp[2];
                for (int w = 0; w < 2; ++w) {
                    cc[w] = two[w] >= 0 ? cost_of[(size_t)two[w]] : ccost[top.order[(size_t)RPTR_BVH_LEAF_FIRST(two[w])]];
                    exp[w] = c_id[w] >= n && c_cnt[w] > (uint32_t)RP_LBVH_LEAF_TRIS;
                }
                auto G = [&](int w, int q) { return !exp[w] ? 0.0f : q == 1 ? cc[w].x : std::fmin(cc[w].x, q == 2 ? cc[w].y : q == 3 ? cc[w].z : cc[w].w); };
                const float f2 = G(0, 1) + G(1, 1), f3 = std::fmin(G(0, 1) + G(1, 2), G(0, 2) + G(1, 1)),
                            f4 = std::fmin(std::fmin(G(0, 1) + G(1, 3), G(0, 2) + G(1, 2)), G(0, 3) + G(1, 1));
                tcost[k] = cost_of[(size_t)i] = make_float4(ta[k] + f4, f2, f3, f4);
            }
            id_of[(size_t)i] = first_id + (uint32_t)k;
            cnt_of[(size_t)i] = tc[k];
        }
        DB_ALLOC(d_tl, int, T);
        DB_ALLOC(d_tr, int, T);
        DB_ALLOC(d_tc, uint32_t, T);
        DB_ALLOC(d_ta, float, T);
        DB_TRY(hipMemcpyAsync(d_ta, ta.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tl, tl.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tr, tr.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(d_tc, tc.data(), T * 4, hipMemcpyHostToDevice, st));
        DB_TRY(hipMemcpyAsync(dpc + (first_id - n), tcost.data(), T * sizeof(float4), hipMemcpyHostToDevice, st)); // (top node k has id first_id + k)
        hipLaunchKernelGGL(rp_k_ploc_stitch, dim3(grid_for(h, T)), dim3(256), 0, st, (uint32_t)T, n, first_id, d_tl, d_tr, d_tc, d_ta, left, right, parent, count, area);
        DB_TRY(hipStreamSynchronize(st)); // (the host arrays are read by the copies above)
        if (first_id + (uint32_t)T != 2u * n - 1u) {
            fail(h, RPTR_E_HIP, "device BVH build: %u + %zu nodes for %u triangles", first_id, T, n);
            return false;
        }
    } else if (host_totals[1] != 2u * n - 1u) {
        fail(h, RPTR_E_HIP, "device BVH build: %u nodes for %u triangles", host_totals[1], n);
        return false;
    }
    out.ms_top = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_top0).count();
    // 5. depth-first order of the triangles (every subtree a contiguous range)
    const uint32_t root_id = 2u * n - 2u;
    DB_ALLOC(nfirst, uint32_t, n);
    hipLaunchKernelGGL(rp_k_ploc_firsts, dim3(g), dim3(256), 0, st, n, left, right, parent, count, nfirst);
    float *tri_box = cbox_b; // (the cluster box lists are dead after the stitch: one of them takes the triangle bounds in their final order)
    hipLaunchKernelGGL(rp_k_ploc_scatter, dim3(g), dim3(256), 0, st, n, left, right, parent, count, tris_b, box_b, tris_a, tri_box);
    // 6. 4-wide collapse, breadth first: a launch pair + one scan per depth level; then boxes + encoding, deepest level first
    DB_ALLOC(nodes, RptrBvh4Node, n);
    DB_ALLOC(node_box, float, 6 * (size_t)n);
    DB_ALLOC(queue_a, int, n);
    DB_ALLOC(queue_b, int, n);
    DB_ALLOC(d_next, uint32_t, 1);
    std::vector<uint32_t> level_base;
    {
        const int h_root = (int)root_id;
        DB_TRY(hipMemcpyAsync(queue_a, &h_root, sizeof(int), hipMemcpyHostToDevice, st));
        uint32_t size = 1, base = 0;
        while (size > 0) {
            if (level_base.size() >= 2 * RP_REFIT_LEVELS || (size_t)base + size > (size_t)n) {
                fail(h, RPTR_E_UNSUPPORTED, "device BVH build: a tree of more than %d levels / %u nodes", 2 * RP_REFIT_LEVELS, base + size);
                return false;
            }
            level_base.push_back(base);
            const int gl = grid_for(h, size);
            hipLaunchKernelGGL(rp_k_ploc_collapse_count, dim3(gl), dim3(256), 0, st, queue_a, size, n, left, right, count, dpc, flag);
            bytes = cub_bytes;
            DB_TRY(hipcub::DeviceScan::ExclusiveSum(cub_tmp, bytes, flag, slot, (int)size, st));
            hipLaunchKernelGGL(rp_k_ploc_collapse_emit, dim3(gl), dim3(256), 0, st, queue_a, size, n, left, right, parent, count, dpc, nfirst, slot, base, base + size, nodes,
                               queue_b, d_next);
            uint32_t next = 0;
            DB_TRY(hipMemcpyAsync(&next, d_next, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            DB_TRY(hipStreamSynchronize(st));
            base += size;
            size = next;
            std::swap(queue_a, queue_b);
        }
        level_base.push_back(base); // = the number of nodes
    }
    const int h_count = (int)level_base.back();
    for (size_t l = level_base.size() - 1; l-- > 0;)
        hipLaunchKernelGGL(rp_k_ploc_refit_range, dim3(grid_for(h, level_base[l + 1] - level_base[l])), dim3(256), 0, st, nodes, node_box, tri_box, level_base[l], level_base[l + 1]);
    (void)hipEventRecord(e1, st);
    // back to the host, in the host builder's form
    out.nodes.resize((size_t)h_count);
    out.boxes.resize((size_t)h_count);
    out.tris.resize(n);
    DB_TRY(hipMemcpyAsync(out.nodes.data(), nodes, (size_t)h_count * sizeof(RptrBvh4Node), hipMemcpyDeviceToHost, st));
    DB_TRY(hipMemcpyAsync(out.boxes.data(), node_box, (size_t)h_count * 24, hipMemcpyDeviceToHost, st));
    DB_TRY(hipMemcpyAsync(out.tris.data(), tris_a, (size_t)n * sizeof(RptrBvhTri), hipMemcpyDeviceToHost, st));
    DB_TRY(hipStreamSynchronize(st));
    DB_TRY(hipGetLastError());
    float ms = 0.f;
    if (e0 && e1 && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) out.ms_device = ms;
    for (RptrBvh4Node &nd : out.nodes) nd._pad1[0] = 0; // (the depth parked there by the emit kernel is not part of the tree)
    return true;
#undef DB_TRY
#undef DB_ALLOC
}

static void build_host_bvh(const RptrSceneDesc *s, HostBvh &B, const RpOptions &opt, const DeviceBuildCtx *dev = nullptr) {
    // instanceCustomIndex of every parameterized mesh = number of geometries before it (render_vulkan.cpp:2748-2850)
    std::vector<int> pmesh_base(s->num_parameterized_meshes, 0);
    {
        int at = 0;
        for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
            pmesh_base[p] = at;
            at += (int)s->meshes[s->parameterized_meshes[p].mesh].num_geometries;
        }
    }
    // ---- bottom-level BVHs (one per mesh), built from dequantised floats
    B.nodes.clear();
    B.tris.clear();
    B.insts.clear();
    B.meshes.assign(s->num_meshes, MeshRt());
    std::vector<RptrBvh4Node> blas_nodes;          // relocated behind the TLAS afterwards
    std::vector<std::array<float, 6>> blas_boxes;  // exact float bounds per node (refit + instance bounds)
    // encodes a wide tree into 64-byte nodes; inner child indices get `node_shift`, leaf ranges `first_shift`
    auto encode_tree = [](const rptr::Wide4Tree &wt, int node_shift, int first_shift, std::vector<RptrBvh4Node> &dst,
                          std::vector<std::array<float, 6>> &boxes) {
        for (const rptr::Wide4 &w : wt.nodes) {
            int32_t child[4];
            for (int k = 0; k < 4; ++k) {
                const int32_t c = w.child[k];
                if (c == RPTR_BVH4_EMPTY)
                    child[k] = c;
                else if (c >= 0)
                    child[k] = c + node_shift;
                else
                    child[k] = RPTR_BVH_LEAF(RPTR_BVH_LEAF_FIRST(c) + first_shift, RPTR_BVH_LEAF_COUNT(c));
            }
            RptrBvh4Node n;
            std::array<float, 6> nb;
            rp_bvh4_encode(w.box, child, &n, nb.data(), nb.data() + 3);
            dst.push_back(n);
            boxes.push_back(nb);
        }
    };
    // candidates of the alpha test: a triangle is flagged when some parameterized mesh of its mesh assigns it a material
    // without BASE_MATERIAL_NOALPHA (the material is per parameterized mesh, the BLAS per mesh; the test itself looks
    // the material up again, kernels.h rp_alpha_rejects)
    std::vector<std::vector<uint8_t>> tri_alpha(s->num_meshes);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        std::vector<uint8_t> &fl = tri_alpha[pm.mesh];
        size_t off = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t nt = s->geometries[mesh.first_geometry + j].num_tris;
            if (fl.size() < off + nt) fl.resize(off + nt, 0);
            for (uint32_t t = 0; t < nt; ++t) {
                const int64_t mid = (int64_t)pm.material_offsets[j] + (pm.tri_material_ids ? (int64_t)pm.tri_material_ids[off + t] : 0);
                if (mid >= 0 && mid < (int64_t)s->num_materials && (s->materials[mid].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0) fl[off + t] = 1;
            }
            off += nt;
        }
    }
    const int flatten_mode = want_flatten(s, opt);
    const bool flatten = flatten_mode != 0, partial = flatten_mode == 2;
    // where a flattened triangle's OWN instance record lies in the instance array: behind the records the top level refers to (one for the
    // flat tree; PARTIAL: + one per instance of a dynamic mesh), at flat_bias + its instance id
    uint32_t n_dynamic_insts = 0;
    auto instance_is_dynamic = [&](uint32_t i) { return s->meshes[s->parameterized_meshes[s->instances[i].parameterized_mesh].mesh].dynamic != 0; };
    for (uint32_t i = 0; i < s->num_instances; ++i) n_dynamic_insts += (partial && instance_is_dynamic(i)) ? 1u : 0u;
    const uint32_t flat_bias = 1u + n_dynamic_insts;
    B.flat_id_bias = flatten ? (int)flat_bias : 0;
    rptr::build_tuning().collapse_rule = (int)opt.v[OPT_COLLAPSE];
    rptr::build_tuning().ploc_top = (size_t)opt.v[OPT_PLOC_TOP];
    rptr::build_tuning().ploc_leaf = (int)opt.v[OPT_PLOC_LEAF];
    // spatial splits for static geometry (bvh_build.h presplit_triangles): RPTR_PRESPLIT="density[,budget]". Off unless asked for:
    // on the 10 M-triangle forest they buy 16 % fewer triangle tests for 7 % more node visits and twice the references
    // (profiles/r03_notes.md), on height fields nothing
    const float split_density = (float)opt.v[OPT_PRESPLIT_DENSITY], split_budget = (float)opt.v[OPT_PRESPLIT_BUDGET_PCT] * 0.01f;
    // who builds a bottom-level tree: option "bvh_builder" = 0 auto (the device for large static triangle sets, the host otherwise), 1 host, 2 device
    const int builder_mode = (int)opt.v[OPT_BVH_BUILDER];
    const int host_ploc = (int)opt.v[OPT_HOST_PLOC];
    auto on_device = [&](size_t n_tris) {
        return dev && dev->build && builder_mode != 1 && n_tris >= 2 && n_tris < ((size_t)1 << 28) && (builder_mode == 2 || n_tris >= dev->min_tris) &&
               !(split_density > 0.0f && split_budget > 0.0f);
    };
    bool flat_done = false;
    if (flatten) { // the one world-space tree on the device: triangles from the vertex streams, sort, clustering, collapse, encoding (ploc.h)
        size_t total = 0;
        std::vector<RpBuildSegment> segs;
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
            const RptrMeshDesc &mesh = s->meshes[pm.mesh];
            if (mesh.dynamic) continue; // (PARTIAL: its instances keep their own records and trees)
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                if (!gd.num_tris) continue;
                RpBuildSegment sg;
                memset(&sg, 0, sizeof(sg));
                if (dev && dev->d_qpos) {
                    sg.qpos = (*dev->d_qpos)[mesh.first_geometry + j];
                    sg.mat_ids = (*dev->geoms)[(size_t)pmesh_base[in.parameterized_mesh] + j].mat_ids;
                }
                memcpy(sg.scaling, gd.quantized_scaling, 12);
                memcpy(sg.offset, gd.quantized_offset, 12);
                sg.material_offset = pm.material_offsets[j];
                sg.begin = (uint32_t)total;
                memcpy(sg.transform, in.transform, 48);
                sg.count = gd.num_tris;
                sg.geom = j;
                sg.flags_hi = (flat_bias + i) << 8;
                sg.has_transform = 1;
                segs.push_back(sg);
                total += gd.num_tris;
            }
        }
        DeviceTree dt;
        if (on_device(total) && dev->build(segs, (uint32_t)total, dt)) {
            B.flat_tris = dt.tris.size();
            B.flat_nodes = dt.nodes.size();
            for (MeshRt &mr : B.meshes) {
                mr.node_base = 0;
                mr.node_count = 0;
                mr.tri_base = 0;
                mr.tri_count = 0;
                memcpy(mr.lo, dt.boxes[0].data(), 12);
                memcpy(mr.hi, dt.boxes[0].data() + 3, 12);
            }
            B.tris = std::move(dt.tris);
            blas_nodes = std::move(dt.nodes);
            blas_boxes = std::move(dt.boxes);
            B.device_built = true;
            B.device_ms = dt.ms_device;
            B.device_iterations = dt.iterations;
            flat_done = true;
        }
    }
    if (flatten && !flat_done) {
        std::vector<rptr::BuildPrim> prims;
        std::vector<RptrBvhTri> mtris;
        std::vector<rptr::TriVerts> verts;
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
            const RptrMeshDesc &mesh = s->meshes[pm.mesh];
            if (mesh.dynamic) continue;
            const float *M = in.transform;
            size_t off = 0;
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                for (uint32_t t = 0; t < gd.num_tris; ++t) {
                    float v[3][3], w[3][3];
                    for (int k = 0; k < 3; ++k) {
                        dequantize_position(gd.qpos[3 * (size_t)t + k], gd.quantized_scaling, gd.quantized_offset, v[k]);
                        for (int r = 0; r < 3; ++r) w[k][r] = ((M[4 * r] * v[k][0] + M[4 * r + 1] * v[k][1]) + M[4 * r + 2] * v[k][2]) + M[4 * r + 3];
                    }
                    RptrBvhTri tri;
                    rptr::BuildPrim bp;
                    for (int k = 0; k < 3; ++k) {
                        tri.v0[k] = w[0][k];
                        tri.e1[k] = w[1][k] - w[0][k];
                        tri.e2[k] = w[2][k] - w[0][k];
                        bp.lo[k] = std::fmin(w[0][k], std::fmin(w[1][k], w[2][k]));
                        bp.hi[k] = std::fmax(w[0][k], std::fmax(w[1][k], w[2][k]));
                    }
                    const int64_t mid = (int64_t)pm.material_offsets[j] + (pm.tri_material_ids ? (int64_t)pm.tri_material_ids[off + t] : 0);
                    const bool alpha = mid >= 0 && mid < (int64_t)s->num_materials && (s->materials[mid].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0;
                    tri.prim = t;
                    tri.geom = j;
                    tri.flags = (alpha ? RPTR_BVH_TRI_ALPHA : 0u) | ((flat_bias + i) << 8); // its instance: record flat_bias + i of the instance array
                    mtris.push_back(tri);
                    prims.push_back(bp);
                    rptr::TriVerts tv;
                    memcpy(tv.v, w, sizeof(tv.v));
                    verts.push_back(tv);
                }
                off += gd.num_tris;
            }
        }
        std::vector<uint32_t> ref_tri; // reference -> triangle (empty: one reference per triangle)
        if (split_density > 0.0f && split_budget > 0.0f) rptr::presplit_triangles(verts.data(), (uint32_t)verts.size(), split_density, split_budget, 256, 0, prims, ref_tri);
        std::vector<rptr::TriVerts>().swap(verts);
        rptr::BuiltTree tree;
        if (host_ploc > 0) // experiment: the clustering of the device builder, stated on the host
            rptr::build_bvh2_ploc(prims.data(), (uint32_t)prims.size(), host_ploc, RPTR_BVH_MAX_LEAF_TRIS, 0, tree);
        else
            rptr::build_bvh2(prims.data(), (uint32_t)prims.size(), RPTR_BVH_MAX_LEAF_TRIS, 48, 0, tree);
        rptr::Wide4Tree wide;
        rptr::collapse_bvh4(tree, wide);
        for (MeshRt &mr : B.meshes) { // no mesh has a tree of its own: they all point at the one tree
            mr.node_base = 0;
            mr.node_count = 0;
            mr.tri_base = 0;
            mr.tri_count = 0;
            memcpy(mr.lo, tree.lo, 12);
            memcpy(mr.hi, tree.hi, 12);
        }
        B.tris.reserve(tree.order.size());
        for (uint32_t id : tree.order) B.tris.push_back(mtris[ref_tri.empty() ? id : ref_tri[id]]);
        encode_tree(wide, 0, 0, blas_nodes, blas_boxes);
        B.flat_tris = B.tris.size();
        B.flat_nodes = blas_nodes.size();
    }
    for (uint32_t m = 0; m < s->num_meshes && (!flatten || partial); ++m) {
        const RptrMeshDesc &mesh = s->meshes[m];
        if (partial && !mesh.dynamic) continue; // (its instances are part of the flat tree)
        {
            size_t total = 0;
            for (uint32_t j = 0; j < mesh.num_geometries; ++j) total += s->geometries[mesh.first_geometry + j].num_tris;
            if (mesh.dynamic == 0 && on_device(total) && dev->d_qpos) {
                std::vector<RpBuildSegment> segs;
                std::vector<size_t> geom_first(mesh.num_geometries, 0);
                size_t at = 0;
                for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                    const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
                    geom_first[j] = at;
                    if (gd.num_tris) {
                        RpBuildSegment sg;
                        memset(&sg, 0, sizeof(sg));
                        sg.qpos = (*dev->d_qpos)[mesh.first_geometry + j];
                        memcpy(sg.scaling, gd.quantized_scaling, 12);
                        memcpy(sg.offset, gd.quantized_offset, 12);
                        sg.material_offset = -1; // (the alpha flag of a mesh's triangle is the OR over its parameterized meshes: set below)
                        sg.begin = (uint32_t)at;
                        sg.count = gd.num_tris;
                        sg.geom = j;
                        segs.push_back(sg);
                    }
                    at += gd.num_tris;
                }
                DeviceTree dt;
                if (dev->build(segs, (uint32_t)total, dt)) {
                    MeshRt &mr = B.meshes[m];
                    mr.dynamic = false;
                    mr.rebuildable = false;
                    mr.node_base = (int)blas_nodes.size();
                    mr.node_count = mr.node_capacity = (int)dt.nodes.size();
                    mr.tri_base = (int)B.tris.size();
                    mr.tri_count = (int)dt.tris.size();
                    memcpy(mr.lo, dt.boxes[0].data(), 12);
                    memcpy(mr.hi, dt.boxes[0].data() + 3, 12);
                    for (RptrBvhTri &t : dt.tris) {
                        const size_t lin = geom_first[t.geom] + t.prim;
                        if (lin < tri_alpha[m].size() && tri_alpha[m][lin]) t.flags |= RPTR_BVH_TRI_ALPHA;
                    }
                    for (RptrBvh4Node &nd : dt.nodes) // local -> absolute references (what encode_tree's shifts do for a host tree)
                        for (int k = 0; k < 4; ++k) {
                            const int32_t c = nd.child[k];
                            if (c == RPTR_BVH4_EMPTY) continue;
                            nd.child[k] = c >= 0 ? c + mr.node_base : RPTR_BVH_LEAF(RPTR_BVH_LEAF_FIRST(c) + mr.tri_base, RPTR_BVH_LEAF_COUNT(c));
                        }
                    B.tris.insert(B.tris.end(), dt.tris.begin(), dt.tris.end());
                    blas_nodes.insert(blas_nodes.end(), dt.nodes.begin(), dt.nodes.end());
                    blas_boxes.insert(blas_boxes.end(), dt.boxes.begin(), dt.boxes.end());
                    B.device_built = true;
                    B.device_ms += dt.ms_device;
                    B.device_iterations = std::max(B.device_iterations, dt.iterations);
                    continue;
                }
            }
        }
        std::vector<rptr::BuildPrim> prims;
        std::vector<RptrBvhTri> mtris;
        std::vector<rptr::TriVerts> verts;
        const bool split_mesh = mesh.dynamic == 0 && split_density > 0.0f && split_budget > 0.0f; // (a refit recomputes boxes from whole triangles)
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const RptrGeometryDesc &gd = s->geometries[mesh.first_geometry + j];
            for (uint32_t t = 0; t < gd.num_tris; ++t) {
                float v[3][3];
                for (int k = 0; k < 3; ++k) dequantize_position(gd.qpos[3 * (size_t)t + k], gd.quantized_scaling, gd.quantized_offset, v[k]);
                RptrBvhTri tri;
                rptr::BuildPrim bp;
                for (int k = 0; k < 3; ++k) {
                    tri.v0[k] = v[0][k];
                    tri.e1[k] = v[1][k] - v[0][k];
                    tri.e2[k] = v[2][k] - v[0][k];
                    bp.lo[k] = std::fmin(v[0][k], std::fmin(v[1][k], v[2][k]));
                    bp.hi[k] = std::fmax(v[0][k], std::fmax(v[1][k], v[2][k]));
                }
                tri.prim = t;
                tri.geom = j;
                tri.flags = (mtris.size() < tri_alpha[m].size() && tri_alpha[m][mtris.size()]) ? RPTR_BVH_TRI_ALPHA : 0u;
                mtris.push_back(tri);
                prims.push_back(bp);
                if (split_mesh) {
                    rptr::TriVerts tv;
                    memcpy(tv.v, v, sizeof(tv.v));
                    verts.push_back(tv);
                }
            }
        }
        MeshRt &mr = B.meshes[m];
        mr.dynamic = mesh.dynamic != 0;
        mr.rebuildable = mr.dynamic && (mesh.dynamic & RPTR_MESH_SUBTLY_DYNAMIC) == 0;
        std::vector<uint32_t> ref_tri;
        if (split_mesh) rptr::presplit_triangles(verts.data(), (uint32_t)verts.size(), split_density, split_budget, 256, 0, prims, ref_tri);
        rptr::BuiltTree tree;
        if (host_ploc > 0) // experiment: the clustering of the device builder, stated on the host
            rptr::build_bvh2_ploc(prims.data(), (uint32_t)prims.size(), host_ploc, RPTR_BVH_MAX_LEAF_TRIS, 0, tree);
        else
            rptr::build_bvh2(prims.data(), (uint32_t)prims.size(), RPTR_BVH_MAX_LEAF_TRIS, 48, 0, tree);
        rptr::Wide4Tree wide;
        // (the meshes of a scene whose instances get several sub-roots each -- partial re-braiding below -- keep the greedy collapse: the cut
        // through the top of the tree wants the balanced nodes it makes; with the area-optimal collapse the instanced forest needs 38.5
        // instead of 36.8 node visits per ray)
        rptr::collapse_bvh4(tree, wide, -1, (s->num_instances >= 16 && host_ploc <= 0) ? rptr::COLLAPSE_GREEDY : rptr::COLLAPSE_OPTIMAL); // (the device builder and its host statement: always the optimal one)
        mr.node_base = (int)blas_nodes.size();
        mr.node_count = (int)wide.nodes.size();
        mr.node_capacity = mr.dynamic ? std::max(mr.node_count, (int)mtris.size()) : mr.node_count;
        mr.tri_base = (int)B.tris.size();
        mr.tri_count = (int)tree.order.size(); // references (= triangles unless the mesh was pre-split)
        memcpy(mr.lo, tree.lo, 12);
        memcpy(mr.hi, tree.hi, 12);
        for (uint32_t id : tree.order) B.tris.push_back(mtris[ref_tri.empty() ? id : ref_tri[id]]);
        encode_tree(wide, mr.node_base, mr.tri_base, blas_nodes, blas_boxes);
        if (mr.node_capacity > mr.node_count) { // room for a device-side rebuild of this dynamic mesh (lbvh.h): unreachable empty nodes
            rptr::Wide4 pad_src;
            (void)pad_src;
            RptrBvh4Node empty;
            memset(&empty, 0, sizeof(empty));
            for (int k = 0; k < 4; ++k) empty.child[k] = RPTR_BVH4_EMPTY;
            blas_nodes.resize((size_t)mr.node_base + mr.node_capacity, empty);
            blas_boxes.resize((size_t)mr.node_base + mr.node_capacity, std::array<float, 6>{0, 0, 0, 0, 0, 0});
        }
    }
    // ---- top level over instance bounds (1 instance record per leaf). Partial re-braiding: when many instances overlap
    // (a forest), one box per instance makes rays enter instance after instance just to leave them at the first nodes.
    // An instance is then represented by up to `braid` records that share transform and ids but start at different
    // sub-roots of its bottom-level tree (the cut is opened largest box first, only through nodes whose children are all
    // inner nodes), each with the world box of its own subtree.
    int braid = s->num_instances >= 16 ? 4 : 1;
    if (opt.v[OPT_REBRAID] > 0) braid = (int)opt.v[OPT_REBRAID];
    if (flatten) braid = 1; // (PARTIAL: the dynamic meshes' instances keep one record each: flat_bias counts on it)
    std::vector<rptr::BuildPrim> iprims;
    std::vector<RptrBvhInstance> insts;
    iprims.reserve((size_t)s->num_instances * braid);
    insts.reserve((size_t)s->num_instances * braid);
    std::vector<std::vector<int>> mesh_cut(B.meshes.size()); // per mesh: the sub-roots (absolute BLAS node indices, before relocation)
    for (size_t m = 0; m < B.meshes.size(); ++m) {
        std::vector<int> cut{B.meshes[m].node_base};
        auto area = [&](int n) {
            const std::array<float, 6> &b = blas_boxes[n];
            const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
            return dx * dy + dy * dz + dz * dx;
        };
        // (a mesh the BVH policy may rebuild keeps its one root: a device-side rebuild gives it a new topology, and sub-roots named by
        // instance records would then point into the middle of another tree; refits keep the topology)
        while ((int)cut.size() < braid && !B.meshes[m].rebuildable) {
            int pick = -1;
            float best = -1.0f;
            for (size_t i = 0; i < cut.size(); ++i) {
                const RptrBvh4Node &nd = blas_nodes[cut[i]];
                int inner = 0, other = 0;
                for (int k = 0; k < 4; ++k) {
                    if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                    (nd.child[k] >= 0 ? inner : other)++;
                }
                if (inner < 2 || other > 0 || (int)cut.size() - 1 + inner > braid) continue; // leaves below it / would overshoot
                const float a = area(cut[i]);
                if (a > best) {
                    best = a;
                    pick = (int)i;
                }
            }
            if (pick < 0) break;
            const RptrBvh4Node nd = blas_nodes[cut[pick]];
            cut.erase(cut.begin() + pick);
            for (int k = 0; k < 4; ++k)
                if (nd.child[k] >= 0) cut.push_back(nd.child[k]);
        }
        mesh_cut[m] = cut;
    }
    std::vector<RptrBvhInstance> own_records; // flattened scene: the scene's instance records, behind the one the top level uses
    if (flatten) {
        RptrBvhInstance bi;
        memset(&bi, 0, sizeof(bi));
        const float identity[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        memcpy(bi.object_to_world, identity, 48);
        memcpy(bi.world_to_object, identity, 48);
        bi.blas_root = 0; // relocated below
        bi.instance_id = -1;
        // (FLAT promises ONE top-level record -- queries start inside it, csrc/dtraverse.h SINGLE, oracle/obvh.h --; the flat tree of a
        // partially flattened scene is an ordinary identity instance of the top level whose triangles name their own records)
        bi.flags = partial ? 0 : RPTR_BVH_INSTANCE_FLAT;
        insts.push_back(bi);
        rptr::BuildPrim bp;
        const std::array<float, 6> &mb = blas_boxes[0];
        for (int k = 0; k < 3; ++k) {
            bp.lo[k] = mb[k];
            bp.hi[k] = mb[3 + k];
        }
        iprims.push_back(bp);
        for (uint32_t i = 0; i < s->num_instances; ++i) {
            const RptrInstanceDesc &in = s->instances[i];
            RptrBvhInstance r;
            memset(&r, 0, sizeof(r));
            memcpy(r.object_to_world, in.transform, 48);
            invert_affine(in.transform, r.world_to_object);
            r.blas_root = -1;
            r.geometry_base = pmesh_base[in.parameterized_mesh];
            r.instance_id = (int)i;
            own_records.push_back(r);
        }
    }
    // the shading records of a mesh's triangles carry the material ids of the FIRST parameterized mesh that uses the mesh (set_scene builds
    // them): instances of any other one resolve theirs through their geometry records
    std::vector<int> first_pmesh_of_mesh(s->num_meshes, -1);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p)
        if (first_pmesh_of_mesh[s->parameterized_meshes[p].mesh] < 0) first_pmesh_of_mesh[s->parameterized_meshes[p].mesh] = (int)p;
    for (uint32_t i = 0; i < s->num_instances && (!flatten || partial); ++i) {
        const RptrInstanceDesc &in = s->instances[i];
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[in.parameterized_mesh];
        if (partial && !s->meshes[pm.mesh].dynamic) continue;
        RptrBvhInstance bi;
        memset(&bi, 0, sizeof(bi));
        memcpy(bi.object_to_world, in.transform, 48);
        invert_affine(in.transform, bi.world_to_object);
        bi.geometry_base = pmesh_base[in.parameterized_mesh];
        bi.instance_id = (int)i;
        if (first_pmesh_of_mesh[pm.mesh] != (int)in.parameterized_mesh) bi.flags |= RPTR_BVH_INSTANCE_OWN_MATERIALS;
        for (int sub : mesh_cut[pm.mesh]) {
            bi.blas_root = sub; // relocated below
            insts.push_back(bi);
            const std::array<float, 6> &mb = blas_boxes[sub]; // exact bounds of the subtree (= the mesh for the root)
            rptr::BuildPrim bp;
            for (int k = 0; k < 3; ++k) {
                bp.lo[k] = INFINITY;
                bp.hi[k] = -INFINITY;
            }
            for (int c = 0; c < 8; ++c) {
                const float p[3] = {c & 1 ? mb[3] : mb[0], c & 2 ? mb[4] : mb[1], c & 4 ? mb[5] : mb[2]};
                const float *M = in.transform;
                for (int r = 0; r < 3; ++r) {
                    const float w = ((M[4 * r] * p[0] + M[4 * r + 1] * p[1]) + M[4 * r + 2] * p[2]) + M[4 * r + 3];
                    bp.lo[r] = std::fmin(bp.lo[r], w);
                    bp.hi[r] = std::fmax(bp.hi[r], w);
                }
            }
            iprims.push_back(bp);
        }
    }
    rptr::BuiltTree tlas;
    rptr::build_bvh2(iprims.data(), (uint32_t)iprims.size(), 1, 24, 1, tlas);
    rptr::Wide4Tree tlas_wide;
    // (the top level keeps the greedy rule: over the heavily overlapping instance boxes of a forest the area-optimal collapse needs 38.5 node
    // visits per ray where the greedy one needs 36.8 -- there the area of a box says little about what a ray does inside it;
    // RPTR_TLAS_COLLAPSE=optimal to try)
    const int tlas_rule = (int)opt.v[OPT_TLAS_COLLAPSE]; // (rptr::COLLAPSE_*: 0 greedy)
    rptr::collapse_bvh4(tlas, tlas_wide, tlas_rule);
    for (int k = 0; k < 3; ++k) {
        B.scene_lo[k] = std::isfinite(tlas.lo[k]) ? tlas.lo[k] : 0.0f;
        B.scene_hi[k] = std::isfinite(tlas.hi[k]) ? tlas.hi[k] : 1.0f;
    }
    const int reloc = (int)tlas_wide.nodes.size();
    B.num_tlas_nodes = reloc;
    B.nodes.clear();
    B.node_box.clear();
    encode_tree(tlas_wide, 0, 0, B.nodes, B.node_box); // TLAS leaf 'first' already indexes the reordered instance array
    for (size_t i = 0; i < blas_nodes.size(); ++i) {
        RptrBvh4Node nd = blas_nodes[i];
        for (int k = 0; k < 4; ++k)
            if (nd.child[k] >= 0) nd.child[k] += reloc;
        B.nodes.push_back(nd);
        B.node_box.push_back(blas_boxes[i]);
    }
    B.mesh_root.assign(B.meshes.size(), -1);
    for (size_t m = 0; m < B.meshes.size(); ++m) {
        B.meshes[m].node_base += reloc;
        B.mesh_root[m] = B.meshes[m].node_base;
    }
    B.insts.resize(insts.size());
    for (size_t k = 0; k < insts.size(); ++k) {
        B.insts[k] = insts[tlas.order[k]];
        B.insts[k].blas_root += reloc;
    }
    B.num_tlas_insts = (int)B.insts.size();
    B.insts.insert(B.insts.end(), own_records.begin(), own_records.end());
    // ---- the traversal stack must hold the worst case of this tree: per node (children - 1) siblings plus whatever
    // its deepest child needs; + the exit marker, + the instance-exit sentinel between the two levels
    {
        const size_t nn = B.nodes.size();
        std::vector<int> need(nn, 0);
        for (int64_t i = (int64_t)nn - 1; i >= 0; --i) { // children sit behind their parents (breadth-first order per tree)
            const RptrBvh4Node &nd = B.nodes[i];
            int nchild = 0, deepest = 0;
            for (int k = 0; k < 4; ++k) {
                if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                ++nchild;
                if (nd.child[k] >= 0) deepest = std::max(deepest, need[nd.child[k]]);
            }
            need[i] = std::max(0, nchild - 1) + deepest;
        }
        int blas_need = 0;
        for (size_t m = 0; m < B.meshes.size(); ++m) blas_need = std::max(blas_need, need[B.mesh_root[m]]);
        const int total = 1 + need[0] + 1 + blas_need;
        B.stack_need = total;
    }
}

static int drain(rptr_hip *h);
static int build_shade_records(rptr_hip *h, SceneCopy &sc, int only_mesh, hipStream_t st);
}

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

int rptr_hip_set_scene(rptr_hip_t *h, const RptrSceneDesc *s) {
    if (!h || !s) return fail(h, RPTR_E_INVALID, "NULL argument");
    HIP_TRY(h, hipSetDevice(h->device));
    {
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (void *p : h->scene_allocs) {
        (void)hipFree(p);
    }
    h->scene_allocs.clear();
    h->bytes_scene = 0;
    h->bytes_allocated = h->bytes_frame;
    h->have_scene = false;
    // ---- validation (what the reference host rejects or this build does not cover yet)
    {
        const std::string bad = validate_scene_tables(s);
        if (!bad.empty()) return fail(h, RPTR_E_INVALID, "%s", bad.c_str());
    }
    if (s->num_textures && !s->textures) return fail(h, RPTR_E_INVALID, "num_textures = %u but textures is NULL", s->num_textures);
    for (uint32_t t = 0; t < s->num_textures; ++t) {
        if (!s->textures[t].rgba8 || s->textures[t].width == 0 || s->textures[t].height == 0 || s->textures[t].width > 16384 || s->textures[t].height > 16384)
            return fail(h, RPTR_E_INVALID, "texture %u: bad size or NULL data", t);
        if (s->textures[t].mip_levels > 1u) { // at most the full chain down to 1 x 1
            uint32_t full = 1;
            for (uint32_t w = s->textures[t].width, hh = s->textures[t].height; w > 1 || hh > 1; w = std::max(1u, w / 2), hh = std::max(1u, hh / 2)) ++full;
            if (s->textures[t].mip_levels > full)
                return fail(h, RPTR_E_INVALID, "texture %u: %u mip levels, a %u x %u texture has at most %u", t, s->textures[t].mip_levels, s->textures[t].width,
                            s->textures[t].height, full);
        }
    }
    h->uses_textures = false;
    h->uses_alpha = false;
    h->tail_adaptive = 1 << 30; // the first frame of a scene shows the queue lengths of every bounce
    for (uint32_t m = 0; m < s->num_materials; ++m) {
        const RptrBaseMaterial &mat = s->materials[m];
        if (mat.normal_map != -1) h->uses_textures = true;
        if (mat.normal_map != -1 && (mat.normal_map < 0 || (uint32_t)mat.normal_map >= s->num_textures))
            return fail(h, RPTR_E_INVALID, "material %u: normal_map %d is not a texture of this scene (%u textures)", m, mat.normal_map, s->num_textures);
        if ((mat.flags & RPTR_BASE_MATERIAL_NOALPHA) == 0) h->uses_alpha = true; // alpha test of hit candidates (kernels.h ALPHA)
        const float vals[5] = {mat.base_color[0], mat.roughness, mat.specular, mat.metallic, mat.ior};
        for (float v : vals) {
            uint32_t u;
            memcpy(&u, &v, 4);
            if (u & RPTR_TEXTURED_PARAM_MASK) h->uses_textures = true;
            if ((u & RPTR_TEXTURED_PARAM_MASK) && RPTR_TEXTURE_ID(u) >= s->num_textures)
                return fail(h, RPTR_E_INVALID, "material %u: textured parameter refers to texture %u of %u", m, RPTR_TEXTURE_ID(u), s->num_textures);
        }
    }
    int rc;
    // ---- textures (RGBA8) + the sRGB decode table
    // paths through a scene with textures carry their texture footprint (kernels.h TEX; the tail kernel's textured instantiation also serves
    // alpha-tested scenes)
    if ((h->uses_textures || h->uses_alpha) && h->path_capacity)
        for (FrameCtx &c : h->ctx)
            if (!c.ps.footprint && (rc = dev_alloc(h, &c.ps.footprint, h->path_capacity, nullptr))) return rc;
    RpTexture *d_textures = nullptr;
    float *d_srgb_lut = nullptr;
    {
        std::vector<RpTexture> tex(s->num_textures);
        for (uint32_t t = 0; t < s->num_textures; ++t) {
            const RptrTextureDesc &td = s->textures[t];
            uchar4 *dt = nullptr;
            const uint32_t levels = td.mip_levels > 1u ? td.mip_levels : 1u;
            size_t n = 0; // the levels back to back, level l = max(1, w >> l) x max(1, h >> l) (vulkan/resource_utils.cpp:86-100)
            for (uint32_t l = 0, w = td.width, hh = td.height; l < levels; ++l, w = std::max(1u, w / 2), hh = std::max(1u, hh / 2)) n += (size_t)w * hh;
            if ((rc = dev_alloc(h, &dt, n, &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(dt, td.rgba8, n * 4, hipMemcpyHostToDevice));
            tex[t].texels = dt;
            tex[t].width = (int)td.width;
            tex[t].height = (int)td.height;
            tex[t].srgb = td.srgb ? 1 : 0;
            tex[t].levels = (int)levels;
        }
        if ((rc = dev_alloc(h, &d_textures, std::max<size_t>(1, tex.size()), &h->scene_allocs))) return rc;
        if (!tex.empty()) HIP_TRY(h, hipMemcpy(d_textures, tex.data(), tex.size() * sizeof(RpTexture), hipMemcpyHostToDevice));
        float lut[256]; // IEC 61966-2-1 decode of an 8-bit code (what a VK_FORMAT_*_SRGB fetch returns before filtering)
        for (int i = 0; i < 256; ++i) {
            const float c = float(i) / 255.0f;
            lut[i] = c <= 0.04045f ? c / 12.92f : std::pow((c + 0.055f) / 1.055f, 2.4f);
        }
        if ((rc = dev_alloc(h, &d_srgb_lut, 256, &h->scene_allocs))) return rc;
        HIP_TRY(h, hipMemcpy(d_srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice));
    }
    // ---- upload vertex streams, one allocation per stream
    std::vector<const uint64_t *> d_qpos(s->num_geometries, nullptr), d_qnu(s->num_geometries, nullptr);
    for (uint32_t g = 0; g < s->num_geometries; ++g) {
        const RptrGeometryDesc &gd = s->geometries[g];
        uint64_t *dp = nullptr;
        if ((rc = dev_alloc(h, &dp, (size_t)gd.num_tris * 3, &h->scene_allocs))) return rc;
        if (gd.num_tris) HIP_TRY(h, hipMemcpy(dp, gd.qpos, (size_t)gd.num_tris * 24, hipMemcpyHostToDevice));
        d_qpos[g] = dp;
        if (gd.qnrm_uv && (gd.has_normals || gd.has_uvs)) {
            uint64_t *dn = nullptr;
            if ((rc = dev_alloc(h, &dn, (size_t)gd.num_tris * 3, &h->scene_allocs))) return rc;
            if (gd.num_tris) HIP_TRY(h, hipMemcpy(dn, gd.qnrm_uv, (size_t)gd.num_tris * 24, hipMemcpyHostToDevice));
            d_qnu[g] = dn;
        }
    }
    // ---- dynamic meshes keep full-precision float positions next to the quantised stream
    h->master.dynpos.assign(s->num_geometries, nullptr);
    h->geom_tris.assign(s->num_geometries, 0);
    h->geom_mesh.assign(s->num_geometries, -1);
    h->master.mesh_dirty.assign(s->num_meshes, 0);
    h->master.mesh_dyn.assign(s->num_meshes, nullptr);
    for (uint32_t m = 0; m < s->num_meshes; ++m) {
        const RptrMeshDesc &mesh = s->meshes[m];
        std::vector<const float *> table(mesh.num_geometries, nullptr);
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t gi = mesh.first_geometry + j;
            const RptrGeometryDesc &gd = s->geometries[gi];
            h->geom_tris[gi] = gd.num_tris;
            h->geom_mesh[gi] = (int)m;
            if (!mesh.dynamic) continue;
            std::vector<float> pos((size_t)gd.num_tris * 9);
            for (size_t v = 0; v < (size_t)gd.num_tris * 3; ++v) dequantize_position(gd.qpos[v], gd.quantized_scaling, gd.quantized_offset, &pos[3 * v]);
            float *dp = nullptr;
            if ((rc = dev_alloc(h, &dp, pos.size(), &h->scene_allocs))) return rc;
            if (!pos.empty()) HIP_TRY(h, hipMemcpy(dp, pos.data(), pos.size() * sizeof(float), hipMemcpyHostToDevice));
            h->master.dynpos[gi] = dp;
            table[j] = dp;
        }
        if (mesh.dynamic) {
            const float **dt = nullptr;
            if ((rc = dev_alloc(h, &dt, table.size(), &h->scene_allocs))) return rc;
            if (!table.empty()) HIP_TRY(h, hipMemcpy(dt, table.data(), table.size() * sizeof(float *), hipMemcpyHostToDevice));
            h->master.mesh_dyn[m] = dt;
            h->master.mesh_dirty[m] = 2;
        }
    }
    // ---- geometry records per (parameterized mesh, geometry): instanced_geometry[] (render_vulkan.cpp:2748-2850)
    std::vector<RpGeomRecord> geoms;
    std::vector<int> pmesh_base(s->num_parameterized_meshes, 0);
    for (uint32_t p = 0; p < s->num_parameterized_meshes; ++p) {
        const RptrParameterizedMeshDesc &pm = s->parameterized_meshes[p];
        const RptrMeshDesc &mesh = s->meshes[pm.mesh];
        pmesh_base[p] = (int)geoms.size();
        size_t total_tris = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) total_tris += s->geometries[mesh.first_geometry + j].num_tris;
        uint8_t *d_ids = nullptr;
        if (pm.tri_material_ids) {
            if ((rc = dev_alloc(h, &d_ids, total_tris, &h->scene_allocs))) return rc;
            if (total_tris) HIP_TRY(h, hipMemcpy(d_ids, pm.tri_material_ids, total_tris, hipMemcpyHostToDevice));
        }
        size_t prim_offset = 0;
        for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
            const uint32_t gi = mesh.first_geometry + j;
            const RptrGeometryDesc &gd = s->geometries[gi];
            RpGeomRecord r;
            memset(&r, 0, sizeof(r));
            r.qpos = d_qpos[gi];
            r.qnrm_uv = d_qnu[gi];
            r.mat_ids = d_ids ? d_ids + prim_offset : nullptr;
            r.dyn_pos = h->master.dynpos[gi];
            memcpy(r.scaling, gd.quantized_scaling, 12);
            memcpy(r.offset, gd.quantized_offset, 12);
            r.material_id = d_ids ? -1 - pm.material_offsets[j] : pm.material_offsets[j];
            r.flags = (gd.has_normals && d_qnu[gi] ? RP_GEOM_HAS_NORMALS : 0u) | (gd.has_uvs && d_qnu[gi] ? RP_GEOM_HAS_UVS : 0u) |
                      (h->master.dynpos[gi] ? RP_GEOM_DYNAMIC : 0u);
            geoms.push_back(r);
            prim_offset += gd.num_tris;
        }
    }
    // ---- acceleration structure (host part, no device involved)
    HostBvh B;
    {
        // large static triangle sets are built on the device (csrc/ploc.h) from the vertex streams uploaded above
        std::vector<uint8_t> mat_alpha(s->num_materials, 0);
        for (uint32_t i = 0; i < s->num_materials; ++i) mat_alpha[i] = (s->materials[i].flags & RPTR_BASE_MATERIAL_NOALPHA) == 0 ? 1 : 0;
        DeviceBuildCtx ctx;
        ctx.d_qpos = &d_qpos;
        ctx.geoms = &geoms;
        ctx.min_tris = (size_t)h->opt.v[OPT_DEVICE_BUILD_MIN_TRIS];
        int device_failures = 0;
        std::string device_failure;
        ctx.build = [&](const std::vector<RpBuildSegment> &segs, uint32_t n, DeviceTree &out) {
            const bool ok = device_build_tree(h, segs, n, mat_alpha, out);
            if (!ok) { // the host builder takes over (same scene, seconds instead of a fraction of one): say so, and do not leave the
                       // message behind as the "last error" of a call that succeeds
                ++device_failures;
                device_failure = h->last_error;
                h->last_error.clear();
            }
            return ok;
        };
        const auto t_build = std::chrono::steady_clock::now();
        build_host_bvh(s, B, h->opt, &ctx);
        if (device_failures && h->opt.v[OPT_QUIET] == 0)
            fprintf(stderr, "rptr_hip: note: %d device-side BVH build(s) failed (%s); the host builder built those trees instead\n", device_failures,
                    device_failure.c_str());
        h->bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
        h->bvh_device_built = B.device_built;
        h->bvh_device_ms = B.device_ms;
    }
    {
        const int capacity = RP_LDS_STACK + RPTR_BVH_STACK_DEPTH;
        if (B.stack_need > capacity)
            return fail(h, RPTR_E_UNSUPPORTED, "the acceleration structure of this scene needs a traversal stack of %d entries (limit %d)",
                        B.stack_need, capacity);
    }
    h->h_nodes = std::move(B.nodes);
    h->h_node_box = std::move(B.node_box);
    h->h_tris = std::move(B.tris);
    h->h_insts = std::move(B.insts);
    h->num_tlas_insts = B.num_tlas_insts;
    h->meshes = std::move(B.meshes);
    h->mesh_root = std::move(B.mesh_root);
    h->num_tlas_nodes = B.num_tlas_nodes;
    h->flat_tris = B.flat_tris;
    h->flat_nodes = B.flat_nodes;
    memcpy(h->scene_lo, B.scene_lo, 12);
    memcpy(h->scene_hi, B.scene_hi, 12);
    // ---- refit: the top level by height (children before parents); the bottom-level trees of dynamic meshes are refitted bottom-up
    // with arrival counters (lbvh.h rp_k_refit_up): per node its parent and the number of its inner children
    std::vector<uint32_t> refit_list;
    h->refit_levels_tlas.clear();
    h->has_dynamic = false;
    for (const MeshRt &mr : h->meshes) h->has_dynamic = h->has_dynamic || mr.dynamic;
    // depth levels of every dynamic mesh's tree (slot RP_REFIT_LEVELS - 1 - depth: ascending slot = deepest first)
    std::vector<uint32_t> h_blas_list(h->h_nodes.size(), 0u);
    std::vector<std::array<uint2, RP_REFIT_LEVELS>> h_levels(h->meshes.size());
    {
        const size_t nn = h->h_nodes.size();
        std::vector<int> height(nn, -1);
        std::vector<std::vector<uint32_t>> tlas_levels;
        // iterative post-order: height = 1 + max(height of inner children), 0 for nodes with leaf children only
        std::vector<std::pair<int, int>> st{{0, 0}};
        while (!st.empty()) {
            auto [n, phase] = st.back();
            st.pop_back();
            const RptrBvh4Node &nd = h->h_nodes[n];
            if (phase == 0) {
                st.push_back({n, 1});
                for (int k = 0; k < 4; ++k)
                    if (nd.child[k] >= 0) st.push_back({nd.child[k], 0});
            } else {
                int hgt = 0;
                for (int k = 0; k < 4; ++k)
                    if (nd.child[k] >= 0) hgt = std::max(hgt, height[nd.child[k]] + 1);
                height[n] = hgt;
                if ((size_t)hgt >= tlas_levels.size()) tlas_levels.resize(hgt + 1);
                tlas_levels[hgt].push_back((uint32_t)n | 0x80000000u);
            }
        }
        for (auto &lv : tlas_levels) {
            h->refit_levels_tlas.push_back({(uint32_t)refit_list.size(), (uint32_t)(refit_list.size() + lv.size())});
            refit_list.insert(refit_list.end(), lv.begin(), lv.end());
        }
        for (size_t m = 0; m < h->meshes.size(); ++m) {
            const MeshRt &mr = h->meshes[m];
            for (auto &l : h_levels[m]) l = make_uint2((uint32_t)mr.node_base, (uint32_t)mr.node_base);
            if (!mr.dynamic) continue;
            std::vector<std::vector<uint32_t>> by_depth;
            std::vector<std::pair<int, int>> bfs{{h->mesh_root[m], 0}};
            for (size_t at = 0; at < bfs.size(); ++at) {
                const auto [n, d] = bfs[at];
                if ((size_t)d >= by_depth.size()) by_depth.resize((size_t)d + 1);
                by_depth[(size_t)d].push_back((uint32_t)n);
                for (int k = 0; k < 4; ++k)
                    if (h->h_nodes[(size_t)n].child[k] >= 0) bfs.push_back({h->h_nodes[(size_t)n].child[k], d + 1});
            }
            if (by_depth.size() > RP_REFIT_LEVELS) return fail(h, RPTR_E_UNSUPPORTED, "mesh %zu: a tree of %zu levels (limit %d)", m, by_depth.size(), RP_REFIT_LEVELS);
            uint32_t at = (uint32_t)mr.node_base;
            for (int slot = 0; slot < RP_REFIT_LEVELS; ++slot) {
                const int d = RP_REFIT_LEVELS - 1 - slot;
                const uint32_t cnt = (size_t)d < by_depth.size() ? (uint32_t)by_depth[(size_t)d].size() : 0u;
                h_levels[m][(size_t)slot] = make_uint2(at, at + cnt);
                for (uint32_t k = 0; k < cnt; ++k) h_blas_list[at + k] = by_depth[(size_t)d][k];
                at += cnt;
            }
        }
    }
    h->rebuild_epoch.assign(h->meshes.size(), 0);
    h->bvh_credit = 0;
    h->rebuild_cursor = 0;
    // ---- upload
    RptrBvh4Node *d_nodes = nullptr;
    RptrBvhTri *d_tris = nullptr;
    RptrBvhInstance *d_insts = nullptr;
    RpGeomRecord *d_geoms = nullptr;
    RptrBaseMaterial *d_mats = nullptr;
    RptrTriLightData *d_lights = nullptr;
    if ((rc = dev_alloc(h, &d_nodes, h->h_nodes.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_tris, h->h_tris.size() + 2, &h->scene_allocs))) return rc; // +2: a leaf is fetched as whole pairs
    if ((rc = dev_alloc(h, &d_insts, h->h_insts.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_geoms, geoms.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &d_mats, s->num_materials, &h->scene_allocs))) return rc;
    // light buffer padded with one zeroed bin (+1): sample_tri_lights may read light_id == bin_end
    const size_t light_cap = (size_t)s->num_lights + RPTR_BINNED_LIGHTS_BIN_MAX_SIZE + 1;
    if ((rc = dev_alloc(h, &d_lights, light_cap, &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &h->d_refit_list, refit_list.size(), &h->scene_allocs))) return rc;
    if ((rc = dev_alloc(h, &h->master.inst_box, (size_t)6 * h->h_insts.size(), &h->scene_allocs))) return rc;
    h->master.tri_box = nullptr;
    if (h->has_dynamic && (rc = dev_alloc(h, &h->master.tri_box, (size_t)6 * h->h_tris.size(), &h->scene_allocs))) return rc;
    if (!refit_list.empty()) HIP_TRY(h, hipMemcpy(h->d_refit_list, refit_list.data(), refit_list.size() * 4, hipMemcpyHostToDevice));
    {
        // the instance bounds and the (small) top-level levels of a refit share one launch (kernels.h rp_k_refit_top): every level is one
        // more dependent launch otherwise, and an animated frame pays for them whatever its size
        const uint32_t small = 4096;
        std::vector<uint2> lv;
        for (auto &l : h->refit_levels_tlas) lv.push_back(make_uint2(l[0], l[1]));
        h->refit_top_all = (size_t)h->num_tlas_insts <= 4 * small; // (only the records the top level refers to have bounds: a flattened tree's triangles name the others)
        for (auto &l : h->refit_levels_tlas) h->refit_top_all = h->refit_top_all && l[1] - l[0] <= small;
        h->d_refit_levels = nullptr;
        if (!lv.empty()) {
            if ((rc = dev_alloc(h, &h->d_refit_levels, lv.size(), &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(h->d_refit_levels, lv.data(), lv.size() * sizeof(uint2), hipMemcpyHostToDevice));
        }
    }
    // level tables + node lists of the dynamic meshes + per-mesh node counts (one set per scene copy: copies are rebuilt independently)
    auto make_refit_tables = [&](SceneCopy &sc) -> int {
        int rc2;
        sc.device_built.assign(h->meshes.size(), 0);
        sc.built_epoch.assign(h->meshes.size(), 0);
        sc.scratch = RpLbvhScratch();
        sc.blas_list = nullptr;
        sc.blas_levels = nullptr;
        sc.mesh_count = nullptr;
        sc.host_levels = h_levels;
        sc.levels_known.assign(h->meshes.size(), 1);
        release_scene_copy_host(sc);
        sc.pinned_levels.assign(h->meshes.size(), nullptr);
        sc.ev_levels.assign(h->meshes.size(), nullptr);
        if (!h->has_dynamic) return RPTR_OK;
        if ((rc2 = dev_alloc(h, &sc.blas_list, h->h_nodes.size(), &h->scene_allocs))) return rc2;
        if ((rc2 = dev_alloc(h, &sc.blas_levels, h->meshes.size() * RP_REFIT_LEVELS, &h->scene_allocs))) return rc2;
        if ((rc2 = dev_alloc(h, &sc.mesh_count, h->meshes.size(), &h->scene_allocs))) return rc2;
        HIP_TRY(h, hipMemcpy(sc.blas_list, h_blas_list.data(), h_blas_list.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(h, hipMemcpy(sc.blas_levels, h_levels.data(), h_levels.size() * sizeof(h_levels[0]), hipMemcpyHostToDevice));
        std::vector<int> counts;
        for (const MeshRt &mr : h->meshes) counts.push_back(mr.node_count);
        HIP_TRY(h, hipMemcpy(sc.mesh_count, counts.data(), counts.size() * sizeof(int), hipMemcpyHostToDevice));
        for (size_t m = 0; m < h->meshes.size(); ++m)
            if (h->meshes[m].dynamic) {
                if (hipHostMalloc((void **)&sc.pinned_levels[m], RP_REFIT_LEVELS * sizeof(uint2), hipHostMallocDefault) != hipSuccess)
                    return fail(h, RPTR_E_NOMEM, "hipHostMalloc failed");
                HIP_TRY(h, hipEventCreateWithFlags(&sc.ev_levels[m], hipEventDisableTiming));
            }
        return RPTR_OK;
    };
    if ((rc = make_refit_tables(h->master))) return rc;
    h->host_bvh_stale = false;
    h->master_refit_pending = false;
    if ((rc = dev_alloc(h, &h->master.node_box, (size_t)6 * h->h_nodes.size(), &h->scene_allocs))) return rc;
    HIP_TRY(h, hipMemcpy(h->master.node_box, h->h_node_box.data(), h->h_node_box.size() * 24, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(d_nodes, h->h_nodes.data(), h->h_nodes.size() * sizeof(RptrBvh4Node), hipMemcpyHostToDevice));
    if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(d_tris, h->h_tris.data(), h->h_tris.size() * sizeof(RptrBvhTri), hipMemcpyHostToDevice));
    if (!h->h_insts.empty())
        HIP_TRY(h, hipMemcpy(d_insts, h->h_insts.data(), h->h_insts.size() * sizeof(RptrBvhInstance), hipMemcpyHostToDevice));
    if (!geoms.empty()) HIP_TRY(h, hipMemcpy(d_geoms, geoms.data(), geoms.size() * sizeof(RpGeomRecord), hipMemcpyHostToDevice));
    if (s->num_materials) HIP_TRY(h, hipMemcpy(d_mats, s->materials, s->num_materials * sizeof(RptrBaseMaterial), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemset(d_lights, 0, light_cap * sizeof(RptrTriLightData)));
    if (s->num_lights) HIP_TRY(h, hipMemcpy(d_lights, s->lights, s->num_lights * sizeof(RptrTriLightData), hipMemcpyHostToDevice));
    h->master.nodes = d_nodes;
    h->master.tris = d_tris;
    h->master.version = h->refit_version;
    h->master.dscene.nodes = d_nodes;
    h->master.dscene.tris = d_tris;
    h->master.dscene.insts = d_insts;
    h->master.dscene.geoms = d_geoms;
    h->master.dscene.materials = d_mats;
    h->master.dscene.lights = d_lights;
    h->master.dscene.num_lights = (int)s->num_lights;
    h->master.dscene.num_materials = (int)s->num_materials;
    h->master.dscene.num_nodes = (uint32_t)h->h_nodes.size();
    h->master.dscene.flat_id_bias = B.flat_id_bias > 0 ? B.flat_id_bias : 1;
    h->master.dscene.single_instance = (h->num_tlas_insts == 1 && h->opt.v[OPT_SINGLE_INSTANCE] != 0) ? 1 : 0;
    h->master.dscene.num_textures = (int)s->num_textures;
    h->master.dscene.textures = d_textures;
    h->master.dscene.srgb_lut = d_srgb_lut;
    // Scheduling thresholds of the traversal (dtraverse.h): a wave refills its idle lanes together once `refill_min` of them have finished,
    // and leaves a node phase for a leaf phase once fewer than `node_min` lanes are at inner nodes. The defaults (48 / 10) were tuned on the
    // height field; in a dense soup of overlapping primitive boxes -- the forest: 26 node visits per ray, node-phase lane utilisation 0.55
    // instead of 0.67, a quarter of the lane slots waiting for a refill -- 32 / 16 are 5 % faster (C4 5.89 -> 5.61 ms) and 1.4 % slower on the
    // height field (profiles/r03_notes.md section 7). The choice follows the tree: the surface-area cost of its largest bottom-level tree
    // (sum of the inner children's box areas over the root's: 11 for the height field, 92 for the flattened forest). RPTR_TRAVERSE_PRESET=
    // "node_min,refill_min" overrides (0,0 = the compile-time defaults).
    {
        double best_cost = 0.0;
        size_t best_tris = 0;
        auto half_area = [&](size_t n) {
            const std::array<float, 6> &b = h->h_node_box[n];
            const double dx = std::max(0.0f, b[3] - b[0]), dy = std::max(0.0f, b[4] - b[1]), dz = std::max(0.0f, b[5] - b[2]);
            return dx * dy + dy * dz + dz * dx;
        };
        for (size_t m = 0; m < h->meshes.size(); ++m) {
            const MeshRt &mr = h->meshes[m];
            // (a mesh without a tree of its own is part of the flattened tree, which lies first: counted once, for mesh 0)
            const size_t root = (size_t)h->mesh_root[m], count = (size_t)(mr.node_count > 0 ? mr.node_count : (m == 0 ? (int)h->flat_nodes : 0));
            const size_t tris_m = mr.tri_count > 0 ? (size_t)mr.tri_count : (m == 0 ? h->flat_tris : 0);
            if (!count || tris_m < best_tris || root >= h->h_nodes.size()) continue;
            const double a0 = half_area(root);
            if (!(a0 > 0.0)) continue;
            double sum = 0.0;
            for (size_t n = root; n < std::min(root + count, h->h_nodes.size()); ++n)
                for (int k = 0; k < 4; ++k)
                    if (h->h_nodes[n].child[k] >= 0) sum += half_area((size_t)h->h_nodes[n].child[k]);
            best_cost = sum / a0;
            best_tris = tris_m;
        }
        // ... times the same measure of the top level (all child boxes of its nodes, instance boxes included, over the scene's box: 1 for a
        // single instance, ~ 6 for the forest's 1001 overlapping instances: the two-level forest gains the same 5 %, 8.38 -> 7.95 ms)
        double tlas_cost = 1.0;
        if (h->num_tlas_nodes > 0 && h->num_tlas_insts > 1) {
            const double a0 = half_area(0);
            double sum = 0.0;
            for (int n = 0; n < h->num_tlas_nodes; ++n) {
                const RptrBvh4Node &nd = h->h_nodes[(size_t)n];
                for (int k = 0; k < 4; ++k) {
                    if (nd.child[k] == RPTR_BVH4_EMPTY) continue;
                    double d[3];
                    for (int a = 0; a < 3; ++a) d[a] = std::max(0.0, (double)((int)nd.qhi[a][k] - (int)nd.qlo[a][k])) * std::ldexp(1.0, (int)nd.exp[a] - 127);
                    sum += d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
                }
            }
            if (a0 > 0.0) tlas_cost = std::max(1.0, sum / a0);
        }
        best_cost *= tlas_cost;
        h->bvh_area_cost = best_cost;
        int node_min = 0, refill_min = 0;
        if (best_cost >= 24.0) { // (height fields: 10-12; the small forests of the tests: 29-35; C4: 90 flattened, 250-280 two-level)
            node_min = 16;
            refill_min = 32;
        }
        if (h->opt.v[OPT_TRAVERSE_NODE_MIN] >= 0) { // options "traverse_node_min" / "traverse_refill_min" (0, 0: the compile-time defaults)
            node_min = (int)h->opt.v[OPT_TRAVERSE_NODE_MIN];
            refill_min = (int)std::max(0ll, h->opt.v[OPT_TRAVERSE_REFILL_MIN]);
        }
        h->master.dscene.node_min = std::max(0, std::min(64, node_min));
        h->master.dscene.refill_min = std::max(0, std::min(64, refill_min));
        h->master.dscene.lds_top = h->opt.v[OPT_LDS_TOP] != 0 ? 1 : 0;
        // ... and the size of a wave's pool of queue entries (dtraverse.h RP_FETCH: 256, four tiles of the first queue): 384 for the trees
        // of the default preset -- one frame at a time 1.82 -> 1.76 ms, two in flight 1.45 -> 1.38 on C2, pipelined unchanged --, 256 for dense
        // ones (the forest loses 4 % with 384; profiles/r05_notes.md section 19)
        h->master.dscene.fetch_max = h->opt.v[OPT_TRAVERSE_FETCH] > 0 ? (int)std::max(64ll, h->opt.v[OPT_TRAVERSE_FETCH] / 64 * 64) : (best_cost >= 24.0 ? 0 : 384);
    }
    // ---- one shading record per BVH triangle (dshade.h RpShadeTri), made on the device from what was just uploaded: per mesh with the
    // geometry records of the first parameterized mesh that uses it, or -- a flattened scene -- per triangle through the instance it names
    {
        RpShadeTri *d_shade = nullptr;
        if ((rc = dev_alloc(h, &d_shade, h->h_tris.size() + 1, &h->scene_allocs))) return rc;
        h->master.shade = d_shade;
        h->master.dscene.shade = d_shade;
        h->mesh_geometry_base.assign(s->num_meshes, -1);
        for (uint32_t p = s->num_parameterized_meshes; p-- > 0;) h->mesh_geometry_base[s->parameterized_meshes[p].mesh] = pmesh_base[p];
        if ((rc = build_shade_records(h, h->master, -1, h->stream))) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->num_lights = (int)s->num_lights;
    h->num_materials = (int)s->num_materials;
    // ---- dynamic scene + frames in flight: every frame context gets its own set of what a refit rewrites
    for (SceneCopy &sc : h->ctx_scene) release_scene_copy_host(sc);
    h->ctx_scene.clear();
    if (h->has_dynamic && h->ctx.size() > 1) {
        h->ctx_scene.resize(h->ctx.size());
        for (SceneCopy &sc : h->ctx_scene) {
            sc.dscene = h->master.dscene;
            sc.mesh_dirty.assign(s->num_meshes, 0);
            sc.dynpos.assign(s->num_geometries, nullptr);
            sc.mesh_dyn.assign(s->num_meshes, nullptr);
            if ((rc = dev_alloc(h, &sc.nodes, h->h_nodes.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.tris, h->h_tris.size() + 2, &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.shade, h->h_tris.size() + 1, &h->scene_allocs))) return rc;
            if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(sc.shade, h->master.shade, h->h_tris.size() * sizeof(RpShadeTri), hipMemcpyDeviceToDevice));
            if ((rc = dev_alloc(h, &sc.node_box, (size_t)6 * h->h_nodes.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.tri_box, (size_t)6 * h->h_tris.size(), &h->scene_allocs))) return rc;
            if ((rc = dev_alloc(h, &sc.inst_box, (size_t)6 * h->h_insts.size(), &h->scene_allocs))) return rc;
            HIP_TRY(h, hipMemcpy(sc.nodes, d_nodes, h->h_nodes.size() * sizeof(RptrBvh4Node), hipMemcpyDeviceToDevice));
            if (!h->h_tris.empty()) HIP_TRY(h, hipMemcpy(sc.tris, d_tris, h->h_tris.size() * sizeof(RptrBvhTri), hipMemcpyDeviceToDevice));
            HIP_TRY(h, hipMemcpy(sc.node_box, h->master.node_box, h->h_node_box.size() * 24, hipMemcpyDeviceToDevice));
            std::vector<RpGeomRecord> cgeoms = geoms; // same records, pointing at this copy's float positions
            for (uint32_t m = 0; m < s->num_meshes; ++m) {
                const RptrMeshDesc &mesh = s->meshes[m];
                if (!mesh.dynamic) continue;
                std::vector<const float *> table(mesh.num_geometries, nullptr);
                for (uint32_t j = 0; j < mesh.num_geometries; ++j) {
                    const uint32_t gi = mesh.first_geometry + j;
                    const size_t nfl = (size_t)s->geometries[gi].num_tris * 9;
                    float *dp = nullptr;
                    if ((rc = dev_alloc(h, &dp, nfl, &h->scene_allocs))) return rc;
                    if (nfl) HIP_TRY(h, hipMemcpy(dp, h->master.dynpos[gi], nfl * sizeof(float), hipMemcpyDeviceToDevice));
                    sc.dynpos[gi] = dp;
                    table[j] = dp;
                }
                const float **dt = nullptr;
                if ((rc = dev_alloc(h, &dt, table.size(), &h->scene_allocs))) return rc;
                if (!table.empty()) HIP_TRY(h, hipMemcpy(dt, table.data(), table.size() * sizeof(float *), hipMemcpyHostToDevice));
                sc.mesh_dyn[m] = dt;
                sc.mesh_dirty[m] = 2;
            }
            for (RpGeomRecord &r : cgeoms)
                if (r.dyn_pos)
                    for (uint32_t gi = 0; gi < s->num_geometries; ++gi)
                        if (r.dyn_pos == h->master.dynpos[gi]) {
                            r.dyn_pos = sc.dynpos[gi];
                            break;
                        }
            RpGeomRecord *cg = nullptr;
            if ((rc = dev_alloc(h, &cg, cgeoms.size(), &h->scene_allocs))) return rc;
            if (!cgeoms.empty()) HIP_TRY(h, hipMemcpy(cg, cgeoms.data(), cgeoms.size() * sizeof(RpGeomRecord), hipMemcpyHostToDevice));
            sc.dscene.nodes = sc.nodes;
            sc.dscene.tris = sc.tris;
            sc.dscene.shade = sc.shade;
            sc.dscene.geoms = cg;
            sc.version = h->refit_version;
            if ((rc = make_refit_tables(sc))) return rc;
        }
    }
    h->have_scene = true;
    // a new scene restarts accumulation (Shell::set_scene -> reset, libapp/shell.cpp:96-126)
    h->frame_offset += h->frame_id;
    h->frame_id = 0;
    return RPTR_OK;
}

static int update_vertices_common(rptr_hip_t *h, uint32_t geometry, const float *xyz, uint32_t num_vertices, bool device_src) {
    if (!h || !xyz) return fail(h, RPTR_E_INVALID, "NULL argument");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "update_vertices before set_scene");
    if (h->ctx_scene.empty()) { // frames in flight read the master vertex buffer and tree
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    if (geometry >= h->master.dynpos.size() || !h->master.dynpos[geometry])
        return fail(h, RPTR_E_INVALID, "geometry %u does not belong to a dynamic mesh (RptrMeshDesc.dynamic)", geometry);
    if (num_vertices != 3u * h->geom_tris[geometry])
        return fail(h, RPTR_E_INVALID, "geometry %u has %u unrolled vertices, got %u", geometry, 3u * h->geom_tris[geometry], num_vertices);
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipMemcpyAsync(h->master.dynpos[geometry], xyz, (size_t)num_vertices * 12, device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                              h->stream));
    if (!device_src) HIP_TRY(h, hipStreamSynchronize(h->stream)); // the host array is only borrowed for the call
    h->master.mesh_dirty[h->geom_mesh[geometry]] = 1;
    h->vertex_updates++;
    return RPTR_OK;
}
int rptr_hip_update_vertices(rptr_hip_t *h, uint32_t geometry, const float *xyz, uint32_t num_vertices) {
    return update_vertices_common(h, geometry, xyz, num_vertices, false);
}
// the reference animates on the device (a compute shader writes float_vertex_buf, render_vulkan.cpp:2834-2840):
// same call with a DEVICE source, ordered on the backend's stream, no host synchronisation
int rptr_hip_update_vertices_device(rptr_hip_t *h, uint32_t geometry, const float *device_xyz, uint32_t num_vertices) {
    return update_vertices_common(h, geometry, device_xyz, num_vertices, true);
}

// ≙ BLAS update (VK_BUILD_ACCELERATION_STRUCTURE_MODE_UPDATE) of the dirty dynamic meshes + TLAS refit
// (render_vulkan.cpp:1323-1354, executed at the top of draw_frame :2165): topology is kept, triangles and all
// boxes are recomputed on the device, level by level from the leaves up.
extern "C++" {
// the shading records (dshade.h RpShadeTri) of one scene copy's triangles: of mesh `only_mesh`, or (-1) of every mesh
static int build_shade_records(rptr_hip *h, SceneCopy &sc, int only_mesh, hipStream_t st) {
    if (h->flat_tris && only_mesh < 0) // the world-space tree over the static instances' triangles: every triangle names its instance record
        hipLaunchKernelGGL(rp_k_build_shade_tris, dim3(grid_for(h, h->flat_tris)), dim3(256), 0, st, sc.dscene, sc.shade, 0u, (uint32_t)h->flat_tris, -1);
    for (size_t m = 0; m < h->meshes.size(); ++m) { // meshes with trees of their own (a mesh inside the flattened tree has none)
        const MeshRt &mr = h->meshes[m];
        if ((only_mesh >= 0 && (int)m != only_mesh) || mr.tri_count <= 0 || h->mesh_geometry_base[m] < 0) continue;
        hipLaunchKernelGGL(rp_k_build_shade_tris, dim3(grid_for(h, (size_t)mr.tri_count)), dim3(256), 0, st, sc.dscene, sc.shade, (uint32_t)mr.tri_base,
                           (uint32_t)mr.tri_count, h->mesh_geometry_base[m]);
    }
    HIP_TRY(h, hipGetLastError());
    return RPTR_OK;
}

// the depth levels of dynamic mesh m of one scene copy, deepest first: a launch per deep level, the shallow ones (at most 4^5 + ... + 1
// nodes) in the single-block kernel, which also does the instance bounds and the top level when `with_top`
static void refit_mesh_levels(rptr_hip *h, SceneCopy &sc, size_t m, hipStream_t st, bool with_top) {
    const MeshRt &mr = h->meshes[m];
    if (!sc.levels_known[m] && sc.ev_levels[m] && hipEventQuery(sc.ev_levels[m]) == hipSuccess) { // the read-back of a device-built tree's table has arrived
        memcpy(sc.host_levels[m].data(), sc.pinned_levels[m], RP_REFIT_LEVELS * sizeof(uint2));
        sc.levels_known[m] = 1;
    }
    const uint2 *dev_levels = sc.blas_levels + m * RP_REFIT_LEVELS;
    const int n_top = 6; // depths 0..5
    for (int slot = 0; slot < RP_REFIT_LEVELS - n_top; ++slot) {
        size_t work = (size_t)mr.node_capacity; // level size unknown to the host: any launch covers it (grid stride)
        if (sc.levels_known[m]) {
            work = sc.host_levels[m][(size_t)slot].y - sc.host_levels[m][(size_t)slot].x;
            if (!work) continue;
        }
        hipLaunchKernelGGL(rp_k_refit_level, dim3(grid_for(h, work, 4)), dim3(256), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.blas_list, dev_levels + slot);
    }
    RptrBvhInstance *insts = const_cast<RptrBvhInstance *>(sc.dscene.insts);
    hipLaunchKernelGGL(rp_k_refit_top, dim3(1), dim3(1024), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box, sc.blas_list,
                       dev_levels + (RP_REFIT_LEVELS - n_top), n_top, h->d_refit_list, h->d_refit_levels, with_top ? (int)h->refit_levels_tlas.size() : 0, insts,
                       with_top ? (uint32_t)h->num_tlas_insts : 0u);
}

// device-side rebuild of the bottom-level tree of dynamic mesh m of one scene copy (lbvh.h), on stream `st`. The triangles of the mesh
// (current order) must hold the new vertices already (rp_k_refit_tris). Ends with the refit that gives the new topology its boxes.
static int lbvh_rebuild(rptr_hip *h, SceneCopy &sc, size_t m, hipStream_t st, bool with_top) {
    const MeshRt &mr = h->meshes[m];
    const uint32_t n = (uint32_t)mr.tri_count;
    RpLbvhScratch &w = sc.scratch;
    if (w.capacity < (size_t)std::max<uint32_t>(n, 2)) { // first rebuild (of a mesh this large): work space for the largest dynamic mesh
        size_t cap = 2;
        for (const MeshRt &x : h->meshes)
            if (x.dynamic) cap = std::max<size_t>(cap, (size_t)x.tri_count);
        // the work space is allocated into a local record and committed as a whole: a failure half way frees what it got (the rebuild is
        // retried with every refit, and a retry must not leak the earlier attempt's buffers while the device is short of memory)
        RpLbvhScratch t = w;
        std::vector<void *> got;
        auto fail_alloc = [&](int code) {
            for (void *p : got) (void)hipFree(p);
            return code;
        };
        auto alloc = [&](auto **out, size_t count) -> int {
            void *p = nullptr;
            const size_t bytes = std::max<size_t>(count, 1) * sizeof(**out);
            hipError_t e = hipMalloc(&p, bytes);
            if (e != hipSuccess) return fail(h, RPTR_E_NOMEM, "hipMalloc(%zu) failed: %s (work space of a device-side BVH rebuild)", bytes, hipGetErrorString(e));
            got.push_back(p);
            *out = reinterpret_cast<std::remove_reference_t<decltype(**out)> *>(p);
            return RPTR_OK;
        };
        int rc;
        if ((rc = alloc(&t.keys_a, cap)) || (rc = alloc(&t.keys_b, cap))) return fail_alloc(rc);
        for (int **p : {&t.left, &t.right, &t.parent, &t.first, &t.last})
            if ((rc = alloc(p, cap))) return fail_alloc(rc);
        for (uint32_t **p : {&t.flag, &t.slot, &t.depth4})
            if ((rc = alloc(p, cap))) return fail_alloc(rc);
        if ((rc = alloc(&t.level_hist, RP_REFIT_LEVELS)) || (rc = alloc(&t.level_cursor, RP_REFIT_LEVELS)) || (rc = alloc(&t.tri_copy, cap)) ||
            (rc = alloc(&t.tribox_copy, 6 * cap)) || (rc = alloc(&t.bounds, 8)))
            return fail_alloc(rc);
        size_t sort_bytes = 0, scan_bytes = 0;
        (void)hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, t.keys_a, t.keys_b, (int)cap, 0, 64, st);
        (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, t.flag, t.slot, (int)cap, st);
        t.cub_bytes = std::max(sort_bytes, scan_bytes) + 256;
        char *tmp = nullptr;
        if ((rc = alloc(&tmp, t.cub_bytes))) return fail_alloc(rc);
        t.cub_tmp = tmp;
        t.capacity = cap;
        for (void *p : got) { // committed: the scene owns the buffers now (an earlier, smaller work space stays until the next set_scene)
            h->scene_allocs.push_back(p);
        }
        w = t;
    }
    RptrBvhTri *tris = sc.tris + mr.tri_base;
    float *tri_box = sc.tri_box + 6ull * mr.tri_base;
    const int g = grid_for(h, n);
    if (n >= 2) {
        hipLaunchKernelGGL(rp_k_lbvh_reset, dim3(1), dim3(64), 0, st, w.bounds);
        hipLaunchKernelGGL(rp_k_lbvh_bounds, dim3(g), dim3(256), 0, st, tri_box, n, w.bounds);
        int index_bits = 1;
        while ((1ull << index_bits) < (unsigned long long)n) ++index_bits;
        hipLaunchKernelGGL(rp_k_lbvh_keys, dim3(g), dim3(256), 0, st, tri_box, n, w.bounds, w.keys_a, index_bits);
        size_t bytes = w.cub_bytes;
        HIP_TRY(h, hipcub::DeviceRadixSort::SortKeys(w.cub_tmp, bytes, w.keys_a, w.keys_b, (int)n, 0, 64, st));
        hipLaunchKernelGGL(rp_k_lbvh_hierarchy, dim3(g), dim3(256), 0, st, w.keys_b, (int)n, w.left, w.right, w.parent, w.first, w.last);
        HIP_TRY(h, hipMemcpyAsync(w.tri_copy, tris, (size_t)n * sizeof(RptrBvhTri), hipMemcpyDeviceToDevice, st));
        HIP_TRY(h, hipMemcpyAsync(w.tribox_copy, tri_box, (size_t)n * 24, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(rp_k_lbvh_gather, dim3(g), dim3(256), 0, st, w.keys_b, n, w.tri_copy, w.tribox_copy, tris, tri_box, (1ull << index_bits) - 1ull);
        hipLaunchKernelGGL(rp_k_lbvh_flags, dim3(g), dim3(256), 0, st, (int)n, w.parent, w.first, w.last, w.flag, w.depth4);
        bytes = w.cub_bytes;
        HIP_TRY(h, hipcub::DeviceScan::ExclusiveSum(w.cub_tmp, bytes, w.flag, w.slot, (int)n - 1, st));
    }
    HIP_TRY(h, hipMemsetAsync(w.level_hist, 0, RP_REFIT_LEVELS * sizeof(uint32_t), st));
    hipLaunchKernelGGL(rp_k_lbvh_emit, dim3(g), dim3(256), 0, st, (int)n, w.left, w.right, w.first, w.last, w.flag, w.slot, w.depth4, mr.node_base, mr.tri_base, sc.nodes,
                       w.level_hist, sc.mesh_count + m);
    uint2 *dev_levels = sc.blas_levels + m * RP_REFIT_LEVELS;
    hipLaunchKernelGGL(rp_k_lbvh_level_scan, dim3(1), dim3(64), 0, st, w.level_hist, (uint32_t)mr.node_base, dev_levels, w.level_cursor);
    hipLaunchKernelGGL(rp_k_lbvh_level_scatter, dim3(grid_for(h, (size_t)mr.node_capacity)), dim3(256), 0, st, sc.nodes, mr.node_base, sc.mesh_count + m, w.level_cursor,
                       sc.blas_list);
    // the host learns the level sizes when this copy has arrived; until then a refit launches every possible level
    sc.levels_known[m] = 0;
    HIP_TRY(h, hipMemcpyAsync(sc.pinned_levels[m], dev_levels, RP_REFIT_LEVELS * sizeof(uint2), hipMemcpyDeviceToHost, st));
    HIP_TRY(h, hipEventRecord(sc.ev_levels[m], st));
    refit_mesh_levels(h, sc, m, st, with_top);
    HIP_TRY(h, hipGetLastError());
    sc.device_built[m] = 1;
    h->rebuilds_done++;
    return RPTR_OK;
}

// refits one copy of the mutable scene on stream `st`; all_dynamic: treat every dynamic mesh as changed. A mesh whose tree is older
// than the rebuild the policy asked for (rptr_hip_refit) is rebuilt instead of refitted. A rebuild that cannot start (no memory for its
// work space) is reported through *err -- the error text is in the handle -- and the mesh is refitted on its old topology instead, so
// that its boxes always match the new vertices; the rebuild is tried again with the next refit.
static bool refit_scene_copy(rptr_hip *h, SceneCopy &sc, bool all_dynamic, hipStream_t st, int *err) {
    bool any = all_dynamic && h->has_dynamic;
    for (size_t m = 0; m < h->meshes.size(); ++m) any = any || sc.mesh_dirty[m] == 1 || (h->meshes[m].dynamic && sc.built_epoch[m] != h->rebuild_epoch[m]);
    if (!any) return false;
    std::vector<size_t> todo;
    for (size_t m = 0; m < h->meshes.size(); ++m) {
        const MeshRt &mr = h->meshes[m];
        if (!mr.dynamic) continue;
        const bool rebuild = sc.built_epoch[m] != h->rebuild_epoch[m];
        if (!all_dynamic && !sc.mesh_dirty[m] && !rebuild) continue; // 1 = new vertices, 2 = dynamic but its triangle bounds were never written
        todo.push_back(m);
    }
    // the instance bounds and the top level ride in the single-block launch of the last mesh when they are small
    bool top_done = false;
    for (size_t k = 0; k < todo.size(); ++k) {
        const size_t m = todo[k];
        const MeshRt &mr = h->meshes[m];
        const bool with_top = h->refit_top_all && k + 1 == todo.size();
        if (mr.tri_count)
            hipLaunchKernelGGL(rp_k_refit_tris, dim3(grid_for(h, (size_t)mr.tri_count)), dim3(256), 0, st, sc.tris, sc.tri_box, sc.shade, (uint32_t)mr.tri_base,
                               (uint32_t)mr.tri_count, sc.mesh_dyn[m]);
        sc.mesh_dirty[m] = 0;
        if (sc.built_epoch[m] != h->rebuild_epoch[m]) {
            const int rc = lbvh_rebuild(h, sc, m, st, with_top);
            if (rc == RPTR_OK) {
                sc.built_epoch[m] = h->rebuild_epoch[m];
                (void)build_shade_records(h, sc, (int)m, st); // the rebuild reordered the mesh's triangles: its shading records follow
            } else {
                if (err && *err == RPTR_OK) *err = rc;
                refit_mesh_levels(h, sc, m, st, with_top);
            }
        } else
            refit_mesh_levels(h, sc, m, st, with_top);
        top_done = top_done || with_top;
    }
    if (!top_done) { // instance bounds, then the top level
        RptrBvhInstance *insts = const_cast<RptrBvhInstance *>(sc.dscene.insts);
        const uint32_t ni = (uint32_t)h->num_tlas_insts;
        if (h->refit_top_all)
            hipLaunchKernelGGL(rp_k_refit_top, dim3(1), dim3(1024), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box, sc.blas_list, sc.blas_levels, 0,
                               h->d_refit_list, h->d_refit_levels, (int)h->refit_levels_tlas.size(), insts, ni);
        else {
            if (ni) hipLaunchKernelGGL(rp_k_refit_instances, dim3(grid_for(h, ni)), dim3(256), 0, st, sc.node_box, insts, sc.inst_box, ni);
            for (auto &lv : h->refit_levels_tlas)
                hipLaunchKernelGGL(rp_k_refit_nodes, dim3(grid_for(h, lv[1] - lv[0])), dim3(256), 0, st, sc.nodes, sc.node_box, sc.tri_box, sc.inst_box,
                                   h->d_refit_list, lv[0], lv[1]);
        }
    }
    return true;
}
} // extern "C++"

int rptr_hip_refit(rptr_hip_t *h) {
    if (!h) return fail(nullptr, RPTR_E_INVALID, "NULL handle");
    if (!h->have_scene) return fail(h, RPTR_E_INVALID, "refit before set_scene");
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->ctx_scene.empty()) { // frames in flight read the master set
        int rc0 = drain(h);
        if (rc0) return rc0;
    }
    // ---- the BVH policy: which dynamic meshes get a new tree instead of a refit (librender/render_params.glsl.h:61,90-93)
    {
        bool changed = false;
        for (size_t m = 0; m < h->meshes.size(); ++m) changed = changed || (h->meshes[m].dynamic && h->master.mesh_dirty[m] == 1);
        if (changed && h->bvh_force_rebuild) {
            for (size_t m = 0; m < h->meshes.size(); ++m)
                if (h->meshes[m].rebuildable && h->master.mesh_dirty[m] == 1) h->rebuild_epoch[m]++;
        } else if (changed && h->bvh_budget > 0) {
            // a budget of triangles per refit call: it is saved up until it covers the next mesh in turn (a mesh larger than the budget is
            // rebuilt every ceil(triangles / budget) calls), dynamic meshes take turns
            long long total = 0;
            std::vector<size_t> dyn;
            for (size_t m = 0; m < h->meshes.size(); ++m)
                if (h->meshes[m].rebuildable) {
                    dyn.push_back(m);
                    total += h->meshes[m].tri_count;
                }
            h->bvh_credit = std::min(h->bvh_credit + h->bvh_budget, std::max(total, h->bvh_budget));
            for (size_t tries = 0; tries < dyn.size() && !dyn.empty(); ++tries) {
                const size_t m = dyn[(size_t)h->rebuild_cursor % dyn.size()];
                if (h->bvh_credit < h->meshes[m].tri_count) break;
                h->bvh_credit -= h->meshes[m].tri_count;
                h->rebuild_epoch[m]++;
                h->rebuild_cursor = (h->rebuild_cursor + 1) % (int)dyn.size();
            }
        }
    }
    if (!h->ctx_scene.empty()) {
        // frames render from the contexts' own sets, which follow from the master's VERTICES when their next frame is submitted:
        // the master's tree is only needed by ray queries and the export, and is refitted when one of them asks for it
        if (h->vertex_updates != h->vertex_updates_refitted) { // (the dirty marks stay for the deferred refit of the master tree)
            h->vertex_updates_refitted = h->vertex_updates;
            h->master_refit_pending = true;
            h->host_bvh_stale = true;
            h->refit_version++;
        }
        return RPTR_OK;
    }
    int err = RPTR_OK;
    if (refit_scene_copy(h, h->master, false, h->stream, &err)) {
        HIP_TRY(h, hipGetLastError());
        h->host_bvh_stale = true;
        h->refit_version++; // the frame contexts' own sets follow when their next frame is submitted
        h->master.version = h->refit_version;
    }
    return err;
}

// the master set's tree after a deferred refit (see rptr_hip_refit)
static int ensure_master_tree(rptr_hip *h) {
    if (!h->master_refit_pending) return RPTR_OK;
    h->master_refit_pending = false;
    int err = RPTR_OK;
    if (refit_scene_copy(h, h->master, false, h->stream, &err)) HIP_TRY(h, hipGetLastError());
    h->master.version = h->refit_version;
    return err;
}

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
static int render_batch_impl(rptr_hip_t *h, const RptrCamera *camera, bool per_frame_cameras, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                             int count_traversal, uint64_t *out_tickets) {
    const int reset_accumulation = reset_first;
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

} // extern "C"

#include "host_comm.h"
