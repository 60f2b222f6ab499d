/*
 * rptr_hip.h -- C ABI of the MI355X-native wavefront path-tracing backend.
 *
 * This is the drop-in boundary for the reference's PT_MEGAKERNEL / RQ_CLOSEST /
 * PROCESS_SAMPLES hot path.  Everything a `RenderBackend` implementation
 * (reference: librender/render_backend.h:68-116) needs from the device side is
 * reachable through the entry points below; signatures carry plain pointers and
 * sizes only.  The reference-side adapter (`RenderHip : RenderBackend`) that a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success or a negative RPTR_E_* code; nothing
 *     throws across the ABI.  rptr_hip_last_error() returns a human-readable
 *     message for the most recent failure on that handle (or the global one
 *     when the handle is NULL).
 *   - all calls for one handle come from one thread (reference:
 *     render_backend.h has no concurrent entry points, SURVEY 8b "Threading").
 *   - host arrays passed to rptr_hip_set_scene are borrowed for the duration
 *     of the call only (reference: Scene is destroyed right after set_scene,
 *     app.cpp:150-175).
 *   - there is NO CPU fallback: without a HIP device every compute entry point
 *     fails with RPTR_E_NO_DEVICE.
 *
 * POD structs are bit-compatible with the reference's shared C++/GLSL structs;
 * the file:line of each is cited next to it.
 */
#ifndef RPTR_HIP_H
#define RPTR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPTR_OK 0
#define RPTR_E_INVALID (-1)    /* bad argument / call order                        */
#define RPTR_E_NO_DEVICE (-2)  /* no HIP device / HIP runtime error                */
#define RPTR_E_NOMEM (-3)
#define RPTR_E_UNSUPPORTED (-4)/* feature of the reference not built yet           */
#define RPTR_E_HIP (-5)        /* a HIP call failed, see rptr_hip_last_error       */

/* ---- compile-time constants of the path (librender/render_params.glsl.h:16-18) */
#define RPTR_MAX_PATH_DEPTH 9
#define RPTR_DEFAULT_RR_PATH_DEPTH 2
#define RPTR_BINNED_LIGHTS_BIN_MAX_SIZE 16
#define RPTR_RAY_EPSILON 0.000005f /* vulkan/gpu_params.glsl:27-29 */

/* material flags, rendering/bsdfs/base_material.h.glsl:7-11. A material WITHOUT NOALPHA is alpha-tested: every hit
 * candidate on it goes through the reference's any-hit test (vulkan/pt_megakernel.glsl:153-212) with the alpha channel of
 * its base colour texture (1 for a literal colour). */
#define RPTR_BASE_MATERIAL_NOALPHA 0x01
#define RPTR_BASE_MATERIAL_ONESIDED 0x02
#define RPTR_BASE_MATERIAL_VOLUME 0x04
#define RPTR_BASE_MATERIAL_EXTENDED 0x08
#define RPTR_BASE_MATERIAL_NEURAL 0x10

/* geometry flags, rendering/rt/geometry.h.glsl:66-70 */
#define RPTR_GEOMETRY_FLAGS_NOALPHA 0x01
#define RPTR_GEOMETRY_FLAGS_IMPLICIT_INDICES 0x02

/* integrator variants of this backend (reference: RenderBackend::variant_names).
 * GLTF   = what PT_MEGAKERNEL ships: glTF metallic-roughness BSDF, 2 lobes
 *          (rendering/bsdfs/gltf_bsdf.glsl, GLTF_SUPPORT_TRANSMISSION off).
 * SIMPLE = Lambert-only material (rendering/bsdfs/simple_bsdf.glsl), the
 *          "diffuse-only BSDF" of BASELINE.json configs[1].                     */
#define RPTR_VARIANT_GLTF 0
#define RPTR_VARIANT_SIMPLE 1
/* GLTF_TRANSMISSION = the same BSDF built with GLTF_SUPPORT_TRANSMISSION + GLTF_SUPPORT_TRANSMISSION_ROUGHNESS (what the reference's
 * MEGAKERNEL_MATERIALS / RT-pipeline hit groups compile, gltf_bsdf.glsl:10-13, vulkan/CMakeLists.txt:35-40): a third lobe for
 * BaseMaterial.specular_transmission -- refraction through BASE_MATERIAL_ONESIDED surfaces, thin "double reflection" through
 * two-sided ones; transmission roughness = roughness, reflection roughness = sqrt(clearcoat_gloss) (gltf_bsdf.glsl:38-62). */
#define RPTR_VARIANT_GLTF_TRANSMISSION 2

/* Point sets: the render backend option rng_variant (librender/render_params.glsl.h:34-37,56,76; rendering/pointsets/selected_rng.glsl).
 * UNIFORM = a per-path LCG (lcg_rng.glsl), the default and what BASELINE.json's configurations use. BN = blue-noise dithered sampling
 * (bn_rng.glsl, BN_OPTIMIZED_SPP 1), SOBOL = the Joe-Kuo Sobol' sequence with a fresh random digital shift per draw (sobol.glsl),
 * Z_SBL = the same sequence with sample indices assigned along a shuffled Z-order curve inside 256 x 256 tiles (Z_ORDER_SHUFFLING). */
#define RPTR_RNG_VARIANT_UNIFORM 0
#define RPTR_RNG_VARIANT_BN 1
#define RPTR_RNG_VARIANT_SOBOL 2
#define RPTR_RNG_VARIANT_Z_SBL 3
/* table sizes in bytes, laid out as the reference uploads them:
 * SobolData (rendering/pointsets/sobol_data.h:13-17): uint32 matrix[1024 * 32], tile_invert_1_0[256 * 256];
 * BNData (bn_data.h:12-27): uint32 sobol_spp_d[256 * 256], tile_scrambling_yx_d_1spp[128 * 128 * 8], then six tables that
 * BN_OPTIMIZED_SPP 1 never reads (they may be left off). */
#define RPTR_SOBOL_TABLE_BYTES ((1024u * 32u + 256u * 256u) * 4u)
#define RPTR_BN_TABLE_MIN_BYTES ((256u * 256u + 128u * 128u * 8u) * 4u)

/* rendering/bsdfs/base_material.h.glsl:13-34 -- 80 bytes */
typedef struct RptrBaseMaterial {
    float base_color[3];
    int32_t normal_map;
    uint32_t flags;
    float roughness;
    float specular;
    float metallic;
    float sheen;
    float sheen_tint;
    float clearcoat;
    float clearcoat_gloss;
    float ior;
    float specular_transmission;
    float anisotropy;
    float specular_tint;
    float transmission_color[3];
    float emission_intensity;
} RptrBaseMaterial;

/* rendering/lights/tri.h.glsl:13-26 -- 48 bytes */
typedef struct RptrTriLightData {
    float v0[3];
    float v1[3];
    float v2[3];
    float radiance[3];
} RptrTriLightData;

/* librender/render_params.glsl.h:165-170 -- 32 bytes */
typedef struct RptrRenderRayQuery {
    float origin[3];
    int32_t mode_or_data;
    float dir[3];
    float t_max;
} RptrRenderRayQuery;

/* librender/render_params.glsl.h:123-128 -- 16 bytes */
typedef struct RptrLightSamplingConfig {
    float light_mis_angle;
    int32_t bin_size;
    float min_perceived_receiver_dist;
    float min_radiance;
} RptrLightSamplingConfig;

/* librender/render_params.glsl.h:130-155 -- 80 bytes */
typedef struct RptrRenderParams {
    int32_t batch_spp;
    int32_t max_path_depth;
    int32_t rr_path_depth;
    int32_t glossy_only_mode;
    float aperture_radius;
    float focus_distance;
    float pixel_radius;
    float variance_radius;
    int32_t output_channel;
    int32_t output_moment;
    float exposure;
    int32_t early_tone_mapping_mode;
    int32_t reprojection_mode;
    int32_t spp_accumulation_window;
    int32_t enable_raster_taa;
    int32_t render_upscale_factor;
    float focal_length;
    int32_t _pad3, _pad4, _pad5;
} RptrRenderParams;

/* rendering/lights/sky_model_arhosek/sky_model.h.glsl:7-10 -- 160 bytes */
typedef struct RptrSkyModelParams {
    float configs[9][4];
    float radiances[4];
} RptrSkyModelParams;

/* the scene-wide constants of vulkan/gpu_params.glsl:120-131 (SceneParams) that
 * the host fills in update_config/update_sky_light (vulkan/render_sky.cpp:25-72).
 * light_count / bin bookkeeping is derived by the backend from set_scene.       */
typedef struct RptrSceneParams {
    RptrSkyModelParams sky_params;
    float sun_dir[3];
    float sun_cos_angle;
    float sun_radiance[4]; /* .w = probability of picking the sun in NEE          */
    float normal_z_scale;
    int32_t _pad[3];
} RptrSceneParams;

/* librender/render_backend.h:26-31 (RenderCameraParams) */
typedef struct RptrCamera {
    float pos[3];
    float dir[3];
    float up[3];
    float fovy; /* degrees */
} RptrCamera;

/* one geometry of a mesh: unrolled, quantized vertex streams
 * (REQUIRE_UNROLLED_VERTICES / QUANTIZED_* of vulkan/gpu_params.glsl:7-9;
 *  stream layout = librender/quantize.h:7-35, librender/dequantize.glsl:8-48). */
typedef struct RptrGeometryDesc {
    const uint64_t *qpos;    /* 3*num_tris, 21 bit/axis                           */
    const uint64_t *qnrm_uv; /* 3*num_tris, lo32 = oct normal, hi32 = uv; or NULL */
    uint32_t num_tris;
    uint32_t has_normals;
    uint32_t has_uvs;
    float quantized_scaling[3];
    float quantized_offset[3];
} RptrGeometryDesc;

/* a mesh = one bottom-level acceleration structure (librender/mesh.h:43-72) */
#define RPTR_MESH_DYNAMIC 1u
#define RPTR_MESH_SUBTLY_DYNAMIC 2u
typedef struct RptrMeshDesc {
    uint32_t first_geometry;
    uint32_t num_geometries;
    uint32_t dynamic; /* Mesh::flags (librender/mesh.h:44-47): 0 static; bit 0 RPTR_MESH_DYNAMIC: vertices may be updated, the tree
                       * is refitted -- or, under rptr_hip_set_bvh_policy, rebuilt on the device; bit 1 RPTR_MESH_SUBTLY_DYNAMIC (small
                       * deformations: SceneLoaderParams::small_deformation): updated and refitted, never rebuilt -- the reference builds
                       * such meshes PREFER_FAST_TRACE | ALLOW_UPDATE instead of PREFER_FAST_BUILD (render_vulkan.cpp:942-952)          */
} RptrMeshDesc;

/* a mesh with a material assignment (librender/mesh.h:78-108, ParameterizedMesh) */
typedef struct RptrParameterizedMeshDesc {
    uint32_t mesh;
    const int32_t *material_offsets; /* one per geometry of the mesh             */
    const uint8_t *tri_material_ids; /* all triangles of the mesh, or NULL       */
} RptrParameterizedMeshDesc;

/* librender/mesh.h:112-116 (Instance) with the transform already dequantized;
 * row-major 3x4 object-to-world like VkAccelerationStructureInstanceKHR
 * (vulkan/render_vulkan.cpp:1262-1268).                                         */
typedef struct RptrInstanceDesc {
    float transform[12];
    uint32_t parameterized_mesh;
} RptrInstanceDesc;

/* A texture: RGBA8, row 0 first, sampled like the reference's material sampler (render_vulkan.cpp:1657-1670: linear filter, linear
 * mip filter, REPEAT addressing, anisotropy 12) over the footprint the path carries to the hit: textureGrad(uv, duvdxy) for material
 * parameters (rendering/rt/material_textures.glsl:37-75 with USE_MIPMAPPING, librender/render_params.glsl.h:8), textureLod(uv, bounce)
 * for normal maps (pt_megakernel.glsl:642-648), level 0 for the any-hit alpha test (:205).
 * mip_levels: 0 or 1 = level 0 only; n = rgba8 holds levels 0..n-1 back to back, level l being max(1, width >> l) x max(1, height >> l)
 * texels (what a .vkt file holds; none is generated: vulkan/resource_utils.cpp:34-100 uploads the levels of the file).
 * srgb != 0: the colour channels are sRGB-encoded (VK_FORMAT_R8G8B8A8_SRGB), alpha is linear. A float material parameter
 * with its sign bit set is a texture handle (rendering/bsdfs/texture_channel_mask.h): bits 0..28 = index into
 * RptrSceneDesc.textures, bits 29..30 = channel for scalar parameters. */
#define RPTR_TEXTURED_PARAM_MASK 0x80000000u
#define RPTR_TEXTURE_ID(bits) ((bits) & 0x1fffffffu)
#define RPTR_TEXTURE_CHANNEL(bits) (((bits) >> 29) & 0x3u)
typedef struct RptrTextureDesc {
    const uint8_t *rgba8;
    uint32_t width, height;
    uint32_t srgb;
    uint32_t mip_levels;
} RptrTextureDesc;

typedef struct RptrSceneDesc {
    const RptrGeometryDesc *geometries;
    uint32_t num_geometries;
    const RptrMeshDesc *meshes;
    uint32_t num_meshes;
    const RptrParameterizedMeshDesc *parameterized_meshes;
    uint32_t num_parameterized_meshes;
    const RptrInstanceDesc *instances;
    uint32_t num_instances;
    const RptrBaseMaterial *materials;
    uint32_t num_materials;
    /* emitters already collected + bin-equalised on the host
     * (librender/lights.cpp:14-90,220-349) */
    const RptrTriLightData *lights;
    uint32_t num_lights;
    /* textures referenced by textured material parameters and normal_map (may be NULL / 0) */
    const RptrTextureDesc *textures;
    uint32_t num_textures;
} RptrSceneDesc;

/* device selection + multi-GPU tile assignment.  The frame is cut
 * into horizontal stripes of `stripe_rows` rows; stripe s belongs to rank
 * s % world_size.  A rank only allocates and renders its own rows. */
/* Threading: a handle is not thread-safe; calls on ONE handle must come from one thread at a time (different handles are
 * independent). Everything is asynchronous to the host only where said so (rptr_hip_render_async, *_device copies). */
/* Version of this header's struct layouts and field meanings. History: 1 = round 1; 2 = RptrTextureDesc._pad became mip_levels, RptrStats grew
 * the stage split; 3 = RptrCreateInfo._pad became abi_version (checked), rank 0 keeps two assembled frames. rptr_hip_abi_version()
 * returns the library's value. Fields named _pad must be zero. 4 = round 5: rptr_hip_set_option / _get_option; rptr_hip_set_frame_schedule /
 * _get_frame_schedule are gone (the device-driven frame schedules they selected lost to the stage launches and left the product).
 * 5 = round 6: RptrCreateInfo.flags (the library no longer touches GPU_MAX_HW_QUEUES unless RPTR_CREATE_SET_HW_QUEUES asks it to),
 * rptr_hip_build_id, option "fast_math", builder experiments behind the "experimental." key prefix. */
#define RPTR_HIP_ABI_VERSION 5
/* RptrCreateInfo.flags */
#define RPTR_CREATE_SET_HW_QUEUES 1u /* rptr_hip_create may set the HIP runtime's GPU_MAX_HW_QUEUES for the WHOLE host process (setenv) when the
                                      * variable is unset or smaller than frames_in_flight + 2 -- effective only when the process has made no HIP
                                      * call yet (the runtime reads the variable once). For hosts that own their process (bin/rptr_hip sets it);
                                      * an embedded plugin leaves it clear and the library never edits its host's environment: it then notes on
                                      * stderr, once, when the frame contexts outnumber the hardware queues (option "quiet" silences it). */
typedef struct RptrCreateInfo {
    int32_t device_ordinal; /* hipSetDevice                                      */
    int32_t rank;
    int32_t world_size;
    int32_t stripe_rows;    /* 0 -> default 32                                   */
    void *stream;           /* hipStream_t to launch on, NULL -> backend-owned   */
    int32_t frames_in_flight; /* 0/1: frames run one after the other on `stream`; 2..16: that many frame contexts with
                               * their own streams, see rptr_hip_render_async. Every context's stream wants a hardware
                               * queue of its own: start the process with GPU_MAX_HW_QUEUES >= frames_in_flight + 1
                               * (HIP runtime variable, default 4; streams that share a queue serialise).          */
    int32_t abi_version;      /* RPTR_HIP_ABI_VERSION of the header the caller was compiled against: rptr_hip_create refuses any other
                               * value (a caller built against an older header would hand over structs whose former padding
                               * fields have since been given a meaning, e.g. RptrTextureDesc.mip_levels) */
    uint32_t flags;           /* RPTR_CREATE_*; 0: nothing outside the handle is touched */
    uint32_t _pad;            /* 0 */
} RptrCreateInfo;

typedef struct RptrStats {
    float render_time_ms;      /* GPU time of the last render call (hipEvents)   */
    float extend_time_ms;      /* sum over closest-hit traversal launches        */
    float connect_time_ms;     /* sum over shadow traversal launches             */
    float shade_time_ms;       /* raygen+sort+shade+resolve                      */
    uint64_t rays_closest;     /* +1 per closest query (pt_megakernel.glsl:440)  */
    uint64_t rays_shadow;      /* +1 per issued shadow query (:223-227)          */
    uint64_t nodes_visited;    /* only when count_traversal was requested: all   */
    uint64_t tris_tested;      /* queries (closest + shadow)                     */
    uint64_t hits_shaded;
    uint64_t nodes_closest;    /* the closest-hit share of nodes_visited         */
    uint64_t tris_closest;
    int32_t spp;               /* accumulated samples per pixel                  */
    int32_t launches_extend;
    int32_t launches_connect;
    int32_t _pad;
    uint64_t device_bytes_allocated;
    /* the parts of shade_time_ms (stage timing level 2): the shade launches alone, the tail kernel (late bounces: extend + shade +
     * connect of a few thousand paths in one launch), the resolve */
    float shade_only_time_ms;
    float tail_time_ms;
    float resolve_time_ms;
    float _pad2;
} RptrStats;

typedef struct rptr_hip rptr_hip_t;

/* ---- lifetime (≙ create_backend_function, render_backend.h:118-119) */
int rptr_hip_create(const RptrCreateInfo *info, rptr_hip_t **out);
int rptr_hip_abi_version(void);
/* a hash over the sources, headers and compiler flags this library was built from (build.py source_id; "unknown" for a hand-made build):
 * measurements that are kept beside the code -- the counter passes of profiles/pmc_traffic.json -- name the build they belong to */
const char *rptr_hip_build_id(void);
/* the acceleration-structure step of the last rptr_hip_set_scene (the reference builds and compacts its BLAS / TLAS on the GPU inside
 * set_scene: vulkan/render_vulkan.cpp:476-543, vulkan/vulkanrt_utils.h:83-105). Large static triangle sets -- a flattened instanced scene,
 * static meshes of millions of triangles -- are built on the device (csrc/ploc.h: Morton sort, PLOC clustering, a binned-SAH top over the
 * remaining clusters, 4-wide collapse, encoding), everything else by the host's binned-SAH builder (options "bvh_builder",
 * "device_build_min_tris"). out_build_ms: wall time of the whole step;
 * out_device_ms: GPU time of the device builds in it (0 when the host built everything). */
int rptr_hip_bvh_build_info(rptr_hip_t *h, int32_t *out_device_built, float *out_build_ms, float *out_device_ms);
/* the scheduling thresholds the traversal kernels use for the current scene (csrc/dtraverse.h: a wave refills its idle lanes once
 * `refill_min` have finished and leaves a node phase once fewer than `node_min` lanes are at inner nodes; 0 = the compile-time defaults
 * 10 / 48) and the measure they were chosen by at set_scene: the surface-area cost of the largest bottom-level tree times that of the top
 * level (>= 24: the dense preset 16 / 32). They change when lanes take their steps, never what a ray finds. Options "traverse_node_min" /
 * "traverse_refill_min" override. No reference counterpart (the reference's traversal is the driver's). */
int rptr_hip_traversal_preset(rptr_hip_t *h, float *out_area_cost, int32_t *out_node_min, int32_t *out_refill_min);
void rptr_hip_destroy(rptr_hip_t *h);
const char *rptr_hip_last_error(const rptr_hip_t *h);
const char *rptr_hip_name(void); /* RenderBackend::name() */
int rptr_hip_set_stream(rptr_hip_t *h, void *hip_stream);

/* ---- RenderBackend::initialize(fb_w, fb_h) (render_vulkan.cpp:246-370) */
int rptr_hip_initialize(rptr_hip_t *h, int fb_width, int fb_height);

/* ---- RenderBackend::set_scene (render_vulkan.cpp:1554-1644): uploads geometry,
 * builds the bottom/top level acceleration structure (include/rptr_bvh.h), uploads materials + binned lights */
int rptr_hip_set_scene(rptr_hip_t *h, const RptrSceneDesc *scene);

/* ---- dynamic meshes: replace the float positions of one geometry and refit
 * (≙ BLAS update + TLAS refit, render_vulkan.cpp:942-952,1323-1354) */
int rptr_hip_update_vertices(rptr_hip_t *h, uint32_t geometry, const float *xyz, uint32_t num_vertices);
/* same with a DEVICE source (the reference animates with a compute shader that writes the float vertex
 * buffer, render_vulkan.cpp:2834-2840): a device-to-device copy ordered on the backend's stream */
int rptr_hip_update_vertices_device(rptr_hip_t *h, uint32_t geometry, const float *device_xyz, uint32_t num_vertices);
int rptr_hip_refit(rptr_hip_t *h);
/* RenderBackendOptions::force_bvh_rebuild / rebuild_triangle_budget (librender/render_params.glsl.h:61,90-93; the reference ships the
 * options and their UI, libapp/app_state.cpp:65-66, but not the animation extension that consumed them): what rptr_hip_refit does with
 * a dynamic mesh whose vertices changed. force_bvh_rebuild != 0: a new tree every time. Otherwise rebuild_triangle_budget triangles
 * may be rebuilt per rptr_hip_refit call -- the budget is saved up until it covers the next dynamic mesh in turn, so a mesh of n
 * triangles gets a new tree every ceil(n / budget) calls and is refitted in between; 0 (the default here) = refit only.
 * A rebuild runs on the device, asynchronously, on the stream of the refit: Morton codes + radix sort + binary radix tree + 4-wide
 * collapse + the encoder of the host builder (csrc/lbvh.h). Ray-query results are those of any tree (closest hit = smallest t, ties
 * by ids); only the number of node visits differs. rptr_hip_bvh_rebuild_count: device-side rebuilds so far (all scene copies). */
int rptr_hip_set_bvh_policy(rptr_hip_t *h, int force_bvh_rebuild, int rebuild_triangle_budget);
int rptr_hip_bvh_rebuild_count(const rptr_hip_t *h, uint64_t *out_rebuilds);

/* ---- RenderBackend::params / lighting_params / update_config
 * (render_backend.h:69-76, render_vulkan.cpp:2943-2959) */
int rptr_hip_set_params(rptr_hip_t *h, const RptrRenderParams *params,
                        const RptrSceneParams *scene_params,
                        const RptrLightSamplingConfig *lighting_params);

/* ---- begin_frame + draw_frame + end_frame for `spp` samples per pixel
 * (render_vulkan.cpp:1919-2178).  Equivalent to `spp` reference frames with
 * batch_spp = 1: sample_index = frame_id .. frame_id+spp-1, each folded into
 * the accumulation buffer by the running mean of process_samples.comp:116-132.
 * reset_accumulation != 0 -> frame_offset += frame_id; frame_id = 0
 * (render_vulkan.cpp:1937-1941).  count_traversal != 0 additionally counts BVH
 * node visits / triangle tests (slower; for roofline accounting only). */
int rptr_hip_render(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp,
                    int reset_accumulation, int count_traversal, RptrStats *out_stats);

/* Frames in flight (≙ the reference's swap-chain frames, `RenderStats.frame_stats_delay`,
 * librender/render_backend.h:15-24): rptr_hip_render_async queues a frame on the next free frame context and
 * returns; rptr_hip_wait blocks until that frame is done and returns its stats. A wavefront frame ends in a
 * latency-bound tail (late bounces carry few rays); with 2-3 frames in flight the tail of one frame overlaps the
 * head of the next. Resolves run in submission order into the one accumulation buffer; read-backs and
 * rptr_hip_copy_tile_to_device return the image of the frame that was waited for last. With frames_in_flight N,
 * at most N tickets can be outstanding (RPTR_E_INVALID otherwise); rptr_hip_render = async + wait, after waiting
 * for everything still in flight, and so do set_scene / update_vertices / refit / trace. */
int rptr_hip_render_async(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp, int reset_accumulation,
                          int count_traversal, uint64_t *out_ticket);
int rptr_hip_wait(rptr_hip_t *h, uint64_t ticket, RptrStats *out_stats);
/* Several frames in ONE launch sequence. A wavefront frame is a chain of dependent launches that each last at least as long as their
 * slowest ray; a small frame (the stripes of one rank of a multi-GPU split) cannot fill the GPU however many frames are in flight. Paths
 * are independent, so the samples of `n_frames` consecutive frames with the same camera and parameters can share the launches: sample
 * slots [k*spp, (k+1)*spp) belong to frame k, which keeps its own frame_offset / sample indices (reset_first: frame 0 restarts the
 * accumulation, reset_rest: so does every further frame -- begin_frame's rule per frame, render_vulkan.cpp:1937-1941), its own image
 * and its own ticket (out_tickets[0..n_frames)). Every frame is bit-identical to the same frame rendered on its own; the resolve folds
 * the frames into the accumulation in order. n_frames * spp sample slots must fit (options "max_batch_spp" /
 * "path_budget_mb": 16 by default), n_frames <= option "max_batch_frames" (8), frames_in_flight >= 2. rptr_hip_wait on any ticket of the batch waits for the batch;
 * read-backs, AOV read-backs (AOV images: of the batch's last frame) and rptr_hip_gather return / send the image of the ticket waited
 * for last; a frame context is free again when all of its batch's tickets have been waited for. RptrStats of a batched frame: an
 * equal share of the batch's times and counts. No reference counterpart (the reference renders one frame per submission). */
int rptr_hip_render_batch_async(rptr_hip_t *h, const RptrCamera *camera, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                                int count_traversal, uint64_t *out_tickets);
/* The same with a camera PER FRAME: cameras[0 .. n_frames) (n_frames <= 8). The reference's loop may move the camera every frame
 * (app.cpp:350-469, vulkan/render_vulkan.cpp:2880-2941: the view parameters are written per frame); primary rays, the texture footprint
 * and the position view of frame k use cameras[k], the AOV images (those of the last frame of the sequence) its view with the view of
 * frame n_frames - 2 as VP_reference. Every frame is bit-identical to the same frame submitted alone with its camera. (Camera rays of
 * such a sequence are made by the kernels' general instantiation -- the one that also serves table-driven point sets.) */
int rptr_hip_render_batch_cameras_async(rptr_hip_t *h, const RptrCamera *cameras, int variant, int spp, int n_frames, int reset_first, int reset_rest,
                                        int count_traversal, uint64_t *out_tickets);
/* RenderConfiguration::freeze_frame (librender/render_backend.h:39; vulkan/render_vulkan.cpp:1937-1941,2152-2154): while set, a reset
 * does not advance frame_offset and a rendered frame does not advance frame_id -- every frame repeats the same samples (the
 * reference's --freeze-frame, cmdline.cpp:359-360). */
int rptr_hip_set_freeze_frame(rptr_hip_t *h, int freeze_frame);
/* set_backend_options(rng_variant) + the upload of the point set's table (vulkan/pointsets/render_sobol.cpp:84-104, render_bn.cpp:78-126:
 * the buffer bound at RANDOM_NUMBERS_BIND_POINT). `table` = host pointer to SobolData / BNData bytes (sizes above; NULL and 0 for
 * UNIFORM); the library keeps a device copy. Frames in flight are waited for; accumulation is not reset (the caller resets, as
 * the reference does when backend options change). A table that is too short, or a variant outside 0..3, fails with RPTR_E_INVALID. */
int rptr_hip_set_rng_variant(rptr_hip_t *h, int rng_variant, const void *table, size_t table_bytes);
/* hipEvent pairs recorded per frame for RptrStats.*_time_ms: 0 none (render_time_ms only; the default since round 5 -- the reference's
 * RenderStats has nothing else), 1 around the closest-hit traversal launches (extend_time_ms), 2 every stage (~0.06 ms per 1080p frame
 * rendered alone). */
int rptr_hip_set_stage_timing(rptr_hip_t *h, int level);

/* ---- Options: how the library builds and schedules, beyond RptrCreateInfo. The reference expresses such intent per mesh and per backend
 * option (Mesh::Dynamic / SubtlyDynamic -> build flags, vulkan/render_vulkan.cpp:942-952; RenderBackendOptions, librender/render_params.glsl.h:
 * 56-93); here every such switch is a named integer. rptr_hip_set_option(h, key, value) changes one for a handle -- it takes effect at the
 * call named below -- and rptr_hip_set_option(NULL, key, value) changes the process default that new handles (and the handle-less
 * rptr_hip_build_bvh_host) start from. Unknown key or value out of range: RPTR_E_INVALID. rptr_hip_get_option reads the value in force.
 * Every option also has an environment variable (right column): the experimenter's override for A/B runs of an unmodified host. When it is set
 * (read once per handle, in rptr_hip_create) its value is in force and rptr_hip_set_option on that key is accepted and ignored. No other
 * environment variable is read by the library (besides RPTR_FRAMES_IN_FLIGHT, which overrides RptrCreateInfo.frames_in_flight, and the HIP
 * runtime's own GPU_MAX_HW_QUEUES, which it reads -- and writes only under RPTR_CREATE_SET_HW_QUEUES). A host that sets NOTHING gets the
 * configuration bench.py measures.
 *
 *   key                      default   takes effect      meaning                                                                   environment
 *   flatten                  -1        set_scene         -1 / 1: a static scene with >= 2 instances (no RPTR_MESH_DYNAMIC mesh) is   RPTR_FLATTEN
 *                                                        built as ONE world-space tree (~150 bytes per instanced triangle; 1.5 x
 *                                                        faster to trace than instance records; hits are found on the pre-transformed
 *                                                        triangles: t / u / v agree with the instance walk to rounding, not bit for
 *                                                        bit); 0: always two-level
 *   flatten_max_tris         1 << 26   set_scene         ... up to this many instanced triangles                                   RPTR_FLATTEN_MAX_TRIS
 *   bvh_builder              0         set_scene         0 auto, 1 host (binned SAH), 2 device (PLOC) for static trees             RPTR_BVH_BUILDER=auto|host|device
 *   device_build_min_tris    2 << 20   set_scene         auto: triangle sets of at least this size are built on the device        RPTR_DEVICE_BUILD_MIN_TRIS
 *   traverse_node_min /      -1        set_scene         scheduling thresholds of the traversal (rptr_hip_traversal_preset);       RPTR_TRAVERSE_PRESET=n,r
 *   traverse_refill_min                                  -1: chosen from the tree
 *   single_instance          1         set_scene         queries of scenes with one instance record start inside it               RPTR_NO_SINGLE_INSTANCE (set = 0)
 *   max_batch_frames         8         initialize        frames a launch sequence may hold (rptr_hip_render_batch_async)           RPTR_MAX_BATCH_FRAMES
 *   max_batch_spp            0         initialize        sample slots in flight per frame context; 0: min(16, what the budget      RPTR_MAX_BATCH_SPP
 *                                                        holds)
 *   path_budget_mb           6144      initialize        path state per frame context                                              RPTR_PATH_BUDGET_MB
 *   blocks_per_cu            0         initialize        persistent traversal blocks per CU; 0: occupancy, shared between the      RPTR_BLOCKS_PER_CU
 *                                                        frame contexts
 *   side_connect             -1        initialize        shadow rays of bounce b on a side stream beside the closest-hit rays of   RPTR_SIDE_CONNECT
 *                                                        b + 1; -1: with one frame context, and with two for a frame submitted
 *                                                        while no other is in flight (a synchronous loop); off with more
 *   aovs                     1         initialize        AOV images (rptr_hip_readback_aov)                                        RPTR_AOVS
 *   tail_bounce              -1        next frame        bounce from which ONE launch finishes the frame; -1 adaptive, 0 never     RPTR_TAIL_BOUNCE
 *   tail_threshold           65536     next frame        adaptive: queue length below which a bounce goes to that launch           RPTR_TAIL_THRESHOLD
 *   stage_timing             0         next frame        = rptr_hip_set_stage_timing                                               RPTR_STAGE_TIMING
 *   comm_transport           0         comm init         0 auto (RCCL between devices), 1 rccl, 2 copy, 3 peer writes              RPTR_COMM_TRANSPORT=rccl|copy|peer
 *   comm_priority            1         comm init         the communication stream has the highest stream priority                 RPTR_COMM_PRIORITY
 *   quiet                    0         -                 no notes on stderr                                                        RPTR_QUIET
 *   traverse_fetch           0         set_scene         queue entries a traversal wave takes per pool at most (multiple of 64);   RPTR_TRAVERSE_FETCH
 *                                                        0: chosen with the thresholds above (384, dense trees 256)
 *   fast_math                0         next frame        the shading stages' division / square root / normalize: 0 IEEE, correctly   RPTR_FAST_MATH
 *                                                        rounded -- the CPU oracle's operations, what every parity test runs on;
 *                                                        1 the hardware's 1-ulp v_rcp / v_sqrt / v_rsq (as a GLSL compiler would:
 *                                                        Vulkan asks 2.5 ulp of `/`): the reference's default renderer (glTF BSDF +
 *                                                        binned-RIS lights) 6.5 % faster per frame, whole frames within 2.3e-4 RMSE of
 *                                                        the oracle (IEEE: 1.4e-5; north_star's tolerance 1e-3), coverage and ray
 *                                                        counts unchanged; csrc/dmath.h. Traversal and camera rays are IEEE either way.
 *   rptr_hip_option_count / rptr_hip_option_name enumerate these keys. Experiments that were measured and not adopted stay reachable for
 *   A/B runs under the key prefix "experimental." (and their environment variables), are not enumerated and carry no promise:
 *   experimental.rebraid, .tlas_collapse, .collapse, .presplit_density, .presplit_budget_pct, .host_ploc, .ploc_top, .ploc_leaf, .lds_top,
 *   .regroup_materials, .comm_self (profiles/r03_notes.md, r05_notes.md say what each lost by).
 *   Read-only through rptr_hip_get_option: "bvh_rebuild_failures" -- device-side rebuilds of dynamic meshes that could not start (no memory for
 *   their work space); such a mesh is refitted on its old topology, the frame is rendered, the next refit tries again; "sample_slots" -- the
 *   sample slots a frame context holds once rptr_hip_initialize has sized the path state ("max_batch_spp", or what "path_budget_mb" allows:
 *   16 up to ~2.9 Mpixel per rank, 12 at 1440p, 5 at 4K): a launch sequence of n frames of s samples needs n * s <= sample_slots. */
int rptr_hip_set_option(rptr_hip_t *h, const char *key, int64_t value);
int rptr_hip_get_option(const rptr_hip_t *h, const char *key, int64_t *out_value);
int rptr_hip_option_count(void);
const char *rptr_hip_option_name(int index);

/* ---- RenderGraphic::get_framebuffer_size / readback_framebuffer
 * (util/display/render_graphic.h:26-37, render_vulkan.cpp:2256-2287).
 * The float read-back returns the RGBA32F accumulation buffer (what
 * --validation writes); rows owned by other ranks are left untouched.
 * Returns the number of floats a full frame needs in *out_count. */
int rptr_hip_get_framebuffer_size(const rptr_hip_t *h, uint32_t out_whc[3]);
int rptr_hip_readback_f32(rptr_hip_t *h, float *rgba, size_t n_floats);
int rptr_hip_readback_u8(rptr_hip_t *h, unsigned char *rgba, size_t n_bytes);
/* RenderGraphic::readback_aov (util/display/render_graphic.h:12-17,40; vulkan/render_vulkan.cpp:2290-2294): the RGBA16F AOV image
 * `aov_index` of the last finished frame, width*height*4 halfs (rows of other ranks stay untouched): 0 albedo.rgb + roughness
 * (1 when ior == 1), 1 shading normal + distance to the camera, 2 screen-space motion.xy + jitter.xy (motion of the hit point
 * between the previous frame's view and this one; no per-vertex motion, jitter 0). Written at bounce 0 by the first sample of a
 * frame (the reference lets every sample of the batch store to the pixel, vulkan/accumulate.glsl:76-103). Option "aovs" = 0 (before
 * rptr_hip_initialize) switches them off. */
#define RPTR_AOV_ALBEDO_ROUGHNESS 0
#define RPTR_AOV_NORMAL_DEPTH 1
#define RPTR_AOV_MOTION_JITTER 2
int rptr_hip_readback_aov(rptr_hip_t *h, int aov_index, uint16_t *rgba16f, size_t n_halfs);

/* ---- multi-GPU: this rank's rows, packed top-to-bottom, for the RCCL gather.
 * rptr_hip_tile_rows writes up to `cap` (first_row,num_rows) pairs for `rank`
 * and returns the number of stripes; rptr_hip_copy_tile_to_device copies this
 * rank's packed rows (float4 per pixel) into a caller-owned DEVICE buffer on
 * the backend's stream. */
int rptr_hip_tile_rows(const rptr_hip_t *h, int rank, int32_t *first_and_count, int cap);
int rptr_hip_local_pixel_count(const rptr_hip_t *h, uint64_t *out_pixels);
int rptr_hip_copy_tile_to_device(rptr_hip_t *h, void *device_dst, size_t n_bytes);

/* ---- multi-GPU: the path's ONE collective, a gather of tile radiance to rank 0 over RCCL / xGMI (north_star; the
 * reference itself is single-GPU: vulkan/render_vulkan_extensions.cpp:77-82 picks one physical device). One communicator rank per
 * handle (RptrCreateInfo.rank / world_size). RCCL is loaded at run time (librccl.so.1, the copy already in the process if there is
 * one), so the library has no link-time dependency on it and single-GPU hosts never touch it.
 *
 *   one process per GPU:   rank 0: rptr_hip_comm_get_unique_id -> hand the 128 bytes to every rank (file, pipe, MPI, torch.distributed)
 *                          every rank: rptr_hip_comm_init_rank(h, id)            (collective: returns when all ranks have joined)
 *                          per frame:  rptr_hip_wait(h, ticket, ..); rptr_hip_gather(h);
 *   one process, n GPUs:   rptr_hip_comm_init_all(handles, n)                    (handles[i] must have rank i, world_size n)
 *                          per frame:  wait every handle's frame; rptr_hip_gather_all(handles, n);
 *
 * rptr_hip_gather sends the packed rows of the frame that was waited for last (the context's image itself: no staging copy) and, on
 * rank 0, receives every other rank's rows and assembles the full frame with one kernel. It is asynchronous: everything runs on
 * a communication stream of its own, ordered behind the waited frame; the next frame that reuses the sending frame context waits
 * for the send on the device, nothing else does -- with frames in flight the gather of frame i overlaps the rendering of frames
 * i+1.. . The assembled frame (rank 0) is read with rptr_hip_readback_gathered_f32 (which waits for the last gather) or used in
 * place through rptr_hip_gathered_frame (valid once rptr_hip_readback_gathered_f32 / rptr_hip_comm_stats have waited for it, or the device has been synchronised by the
 * caller). Rank 0 keeps TWO receive buffers and TWO assembled frames and uses them in turn: the frame of gather g stays intact while
 * gather g + 1 arrives and is assembled (a reader can hold frame i while i + 1 is in flight), and is rewritten by gather g + 2.
 * Handles that share a device (test rigs) and option "comm_transport" = 2 (copy) use peer-to-peer
 * copies (hipMemcpyPeerAsync) instead of RCCL in the one-process mode. "comm_transport" = 3 (peer; one-process mode, devices with peer
 * access): every rank writes its rows straight into their places in rank 0's frame from a kernel on its own communication stream --
 * rank 0 runs no receive kernels and no assembly pass (csrc/host_comm.h "Peer writes"). rptr_hip_comm_transport names what a handle's
 * communicator uses: "rccl", "copy" or "peer" (NULL without a communicator). */
#define RPTR_COMM_ID_BYTES 128
int rptr_hip_comm_get_unique_id(void *out_id128);
int rptr_hip_comm_init_rank(rptr_hip_t *h, const void *id128);
int rptr_hip_comm_init_all(rptr_hip_t *const *handles, int n);
int rptr_hip_comm_destroy(rptr_hip_t *h);
int rptr_hip_gather(rptr_hip_t *h);
int rptr_hip_gather_all(rptr_hip_t *const *handles, int n);
/* ONE collective for the frames of a launch sequence (rptr_hip_render_batch_async / _batch_cameras_async: they finish together and lie
 * behind each other in the frame context's images): call after waiting for the LAST ticket of the sequence; the last n_frames frames of
 * it (1 <= n_frames <= frames per launch sequence; every rank passes the same number) travel in one transfer per rank and are assembled
 * in one pass -- a quarter of the per-frame cost for sequences of four (csrc/host_comm.h "Batched gathers"). n_frames = 1 is
 * rptr_hip_gather. Rank 0 then holds n_frames assembled images: rptr_hip_gathered_frame / rptr_hip_readback_gathered_f32 show the last
 * one, rptr_hip_readback_gathered_frame_f32(h, k, ..) image k (0 = the oldest) of the last gather. */
int rptr_hip_gather_batch(rptr_hip_t *h, int n_frames);
int rptr_hip_gather_all_batch(rptr_hip_t *const *handles, int n, int n_frames);
int rptr_hip_gathered_frame(rptr_hip_t *h, const void **out_device_rgba32f);
int rptr_hip_readback_gathered_f32(rptr_hip_t *h, float *rgba, size_t n_floats);
int rptr_hip_readback_gathered_frame_f32(rptr_hip_t *h, int index, float *rgba, size_t n_floats);
/* gathers issued so far, and the mean GPU time of the completed ones on this rank's communication stream (send / receive + assembly) */
int rptr_hip_comm_stats(rptr_hip_t *h, uint64_t *out_gathers, float *out_mean_gather_ms);
const char *rptr_hip_comm_transport(rptr_hip_t *h);
/* Peer writes for ONE PROCESS PER GPU (opt-in; csrc/host_comm.h COMM_IPC): rank 0 calls rptr_hip_comm_ipc_export (which makes its
 * communicator and writes RPTR_COMM_IPC_BYTES describing its frame buffers: hipIpcGetMemHandle), the bytes reach every rank through any side
 * channel, every rank calls rptr_hip_comm_ipc_init(h, bytes) (rank 0 too: a check). rptr_hip_gather / _gather_batch then scatter every
 * rank's rows straight into rank 0's frame from a kernel on the rank's own communication stream -- no RCCL, no receive buffer, no
 * assembly pass; the ordering between the processes is carried by counters in a flag block of rank 0 (events do not cross processes).
 * Needs HSA_ENABLE_IPC_MODE_LEGACY=0 where the driver only supports dmabuf IPC. rptr_hip_comm_transport says "ipc". */
#define RPTR_COMM_IPC_BYTES 256
int rptr_hip_comm_ipc_export(rptr_hip_t *h, void *out_bytes);
int rptr_hip_comm_ipc_init(rptr_hip_t *h, const void *bytes);

/* ---- enable_ray_queries / render_ray_queries with the RQ_CLOSEST kernel
 * (render_backend.h:101-102, vulkan/rt_intersect.comp:31-68): n queries ->
 * n x float4 (bary.x, bary.y, bits(instance_custom_index + geometry_index),
 * bits(primitive_index)); miss = (-1,-1,bits(-1),bits(-1)); mode_or_data < 0
 * leaves the result slot untouched. Host pointers. */
int rptr_hip_trace(rptr_hip_t *h, const RptrRenderRayQuery *queries, int n, float *out4);
/* The same over DEVICE buffers, asynchronously on `hip_stream` (NULL: the backend's stream) -- what RenderBackend::render_ray_queries works
 * on (vulkan/render_vulkan.cpp:1867-1876: the queries are in ray_query_buffer, written by other device code, the results go to
 * ray_result_buffer). A caller's stream is ordered behind the scene uploads queued on the backend's stream and the backend's later work
 * behind the queries. */
int rptr_hip_trace_device(rptr_hip_t *h, const RptrRenderRayQuery *device_queries, int n, float *device_out4, void *hip_stream);
/* RenderBackend::enable_ray_queries(max_queries, max_queries_per_pixel) (render_backend.h:101, render_vulkan.cpp:430-455): the library
 * owns a device buffer of max(max_queries, width x height x max_queries_per_pixel) RptrRenderRayQuery records and one of as many float4
 * results and returns their device addresses (≙ ray_query_buffer / ray_result_buffer; they stay valid until the next call that asks for
 * more, or rptr_hip_destroy). RenderBackend::render_ray_queries(num_queries, ...) = rptr_hip_render_ray_queries: traces the first
 * num_queries records of that buffer on the backend's stream. */
int rptr_hip_enable_ray_queries(rptr_hip_t *h, int max_queries, int max_queries_per_pixel, void **out_device_queries, void **out_device_results);
int rptr_hip_render_ray_queries(rptr_hip_t *h, int num_queries);
/* RenderBackendOptions::light_sampling_variant (rendering/mc/light_sampling.h:11-20, rendering/mc/nee.glsl:12-14): 0 =
 * LIGHT_SAMPLING_VARIANT_NONE disables next-event estimation towards the emissive triangles (every NEE sample goes to the sun; emitters
 * that a path HITS still contribute, at full weight), 1 = LIGHT_SAMPLING_VARIANT_RIS (the default: binned RIS). The image is that of
 * the scene handed over without its light array and with sun_radiance.w = 1 (tested bit for bit); the lights stay uploaded and come back
 * with variant 1. */
int rptr_hip_set_light_sampling_variant(rptr_hip_t *h, int variant);
/* diagnostic twin: also returns, per query, the node and triangle visits of the traversal
 * (visits2[2*i], visits2[2*i+1]; may be NULL) -- the counts behind the roofline's algorithmic bytes,
 * checked ray by ray against the oracle walking the exported tree. tmin (NULL = the RQ_CLOSEST rule
 * eps*|origin|) gives explicit interval starts; any_hit != 0 runs the occlusion traversal of the
 * shadow rays instead (out4[4*i] = 1 if anything is hit in (tmin, t_max)). No reference counterpart. */
int rptr_hip_trace_counted(rptr_hip_t *h, const RptrRenderRayQuery *queries, int n, float *out4, uint32_t *visits2,
                           const float *tmin, int any_hit);

/* ---- test/diagnostic access to the acceleration structure (node format in
 * DESIGN.md): copies out the flattened BVH so the oracle can traverse the very
 * same tree and count the very same node visits. Pass NULL buffers to query
 * sizes. */
int rptr_hip_export_bvh(rptr_hip_t *h, void *nodes, size_t *n_nodes, void *tris, size_t *n_tris,
                        void *instances, size_t *n_instances);

/* ---- the host half of set_scene on its own: builds the same acceleration structure (binned SAH per mesh, top
 * level over the instance bounds, 4-wide collapse, 64-byte encoding) WITHOUT a device or a handle. Two calls like
 * rptr_hip_export_bvh (NULL buffers to query sizes). out_stack_need: worst-case traversal stack entries of the tree.
 * Test/diagnostic entry: the CPU test suite walks this tree with the oracle. */
int rptr_hip_build_bvh_host(const RptrSceneDesc *scene, void *nodes, size_t *n_nodes, void *tris, size_t *n_tris,
                            void *instances, size_t *n_instances, int32_t *out_stack_need);

/* ---- stats() (render_vulkan.cpp:2229-2243) */
int rptr_hip_stats(const rptr_hip_t *h, RptrStats *out);

#ifdef __cplusplus
}
#endif
#endif /* RPTR_HIP_H */
