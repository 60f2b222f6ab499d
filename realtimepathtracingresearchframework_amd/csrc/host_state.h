// host_state.h -- what every part of the host runtime shares: the handle (rptr_hip), frame contexts, scene copies, the option table and its
// process defaults, error reporting, device allocation, the hardware-queue note
// Part of the ONE translation unit rptr_hip.hip (included there, in this order: host_state.h, host_bvh.inl, host_scene.inl,
// host_frame.inl, host_access.inl, host_comm.h): the host runtime split along its seams; no symbol changed.
#pragma once
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

