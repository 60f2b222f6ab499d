// dshade.h -- device-side shading library of the wavefront path tracer.
//
// HIP restatement (gfx950, wave64) of the parts of the reference's shader
// library that PT_MEGAKERNEL executes per path vertex. Reference locations are
// cited per function (paths relative to the reference checkout). Differences
// from the GLSL text that are deliberate:
//   * GLTF_SUPPORT_TRANSMISSION is off (as in the shipped PT_MEGAKERNEL build),
//     so the component sampler has 2 lobes;
//   * standard textures are the 1x1 texels holding the literal BaseMaterial
//     values (no texture units on this path yet -> normal_map must be -1);
//   * pow(x,5) is ((x*x)*(x*x))*x and pow(x,1.5) is x*sqrt(x).
#pragma once
#include "../../include/rptr_bvh.h"
#include "../../include/rptr_hip.h"
#include "dmath.h"

#define RP_PI 3.14159265358979323846f
#define RP_1_PI 0.318309886183790671538f
#define RP_EPSILON 0.0001f // rendering/defaults.glsl:14-16

// ------------------------------------------------------------------ device scene tables
#define RP_GEOM_HAS_NORMALS 1u
#define RP_GEOM_HAS_UVS 2u
#define RP_GEOM_DYNAMIC 4u

// ≙ RenderMeshParams (rendering/rt/geometry.h.glsl:72-91), one per (parameterized mesh, geometry)
struct RpGeomRecord { // 64 bytes
    const uint64_t *qpos;
    const uint64_t *qnrm_uv;
    const uint8_t *mat_ids; // per-triangle ids of this geometry or NULL
    const float *dyn_pos;   // float positions of dynamic meshes (GEOMETRY_FLAGS_DYNAMIC) or NULL
    float scaling[3];
    int32_t material_id; // >= 0, or -1-offset when mat_ids is used (render_vulkan.cpp:2812-2815)
    float offset[3];
    uint32_t flags;
};

// One shading record per BVH triangle, in the order of RpScene::tris (= leaf order: rays that hit neighbouring triangles read neighbouring
// records), 64 bytes on a 64-byte boundary: what a hit needs of its triangle in ONE cache line, four 16-byte loads that depend on
// nothing but the hit record. Round 4's shade kernels went hit record -> instance record -> geometry record -> qpos[3] + qnrm_uv[3] + a
// material byte (three arrays of 24-byte records that straddle 64-byte lines two times in eight: ~3.5 lines for 49 bytes behind two dependent
// round trips) and waited on memory for 68 % of their wave-cycles (profiles/r04m_*). The record holds
//   pos      the three vertices AS rp_geom_tri RETURNS THEM: float(q) * scaling + offset, evaluated once by the same device function
//            (hit.glsl:41-47, dequantize.glsl:8-21: the same IEEE operations on the same operands -- the shaded values keep their bits); for a
//            dynamic mesh its float positions, rewritten by the refit (kernels_misc.h rp_k_refit_tris)
//   qnu      the three qnrm_uv words (oct normal | uv), decoded per hit as before (dequantize.glsl:23-48)
//   material bits 0..27 the material id (hit.glsl:49-56, resolved for the parameterized mesh the record was made for: an instance of
//            ANOTHER parameterized mesh of the same mesh is flagged RP_INST_OWN_MATERIALS and resolves it through its geometry record),
//            bit 30 / 31 the geometry has normals / uvs
// Built on the device after the acceleration structure (kernels_misc.h rp_k_build_shade_tris) and again for a mesh whose tree was rebuilt.
struct alignas(64) RpShadeTri {
    float pos[9];
    uint32_t qnu[6];
    uint32_t material;
};
#define RP_SHADE_MATERIAL_MASK 0x0FFFFFFFu
#define RP_SHADE_HAS_NORMALS 0x40000000u
#define RP_SHADE_HAS_UVS 0x80000000u
#define RP_INST_OWN_MATERIALS RPTR_BVH_INSTANCE_OWN_MATERIALS

struct RpTexture { // RptrTextureDesc on the device
    const uchar4 *texels;
    int width, height;
    int srgb;
    int levels; // mip levels stored back to back behind level 0 (>= 1)
};

struct RpScene {
    const RptrBvh4Node *nodes;
    const RptrBvhTri *tris;
    const RptrBvhInstance *insts;
    const RpGeomRecord *geoms;
    const RpShadeTri *shade; // one per entry of `tris`
    const RptrBaseMaterial *materials;
    const RptrTriLightData *lights; // padded with one zeroed bin
    int32_t num_lights;
    int32_t num_materials;
    uint32_t num_nodes;
    int32_t single_instance; // the top level holds one instance record: queries start inside it (dtraverse.h)
    int32_t num_textures;
    const RpTexture *textures;
    const float *srgb_lut; // 256 entries: sRGB-encoded byte -> linear float (computed on the host)
    int32_t node_min, refill_min; // scheduling thresholds of the traversal for this scene (dtraverse.h), 0 = the compile-time defaults
    int32_t lds_top;      // option "lds_top": launch the traversal instantiations that stage the top of the tree in LDS (k_extend.hip)
    int32_t flat_id_bias; // a world-space triangle of a flattened tree names instance record r = bias + instance id (1; partially flattened
                          // scenes: 1 + the number of instance records of the dynamic meshes)
    int32_t fetch_max;    // option "traverse_fetch": queue entries a traversal wave takes per pool at most (dtraverse.h RP_FETCH), 0 = the compile-time default
    int32_t _pad_fetch;
};

// Division of a 31-bit number by a frame constant (tiles per row, rows per stripe, padded pixels per sample slot) without the ~25
// instructions of a 32-bit integer division: q = mulhi(n, mul) >> shift with mul = ceil(2^(32 + shift) / d), shift = ceil(log2 d) - 1,
// exact for every n < 2^31 (n * (mul * d - 2^(32+shift)) < 2^(32+shift) because mul * d - 2^(32+shift) < d <= 2^(shift+1)). mul == 0: d == 1.
struct RpDivU32 {
    uint32_t mul, shift;
};
#ifdef __HIPCC__
RP_DEV uint32_t rp_div(uint32_t n, RpDivU32 d) { return d.mul ? (__umulhi(n, d.mul) >> d.shift) : n; }
#endif
static inline RpDivU32 rp_make_div(uint32_t d) { // host side (rptr_hip.hip); d >= 1
    RpDivU32 r{0u, 0u};
    if (d <= 1u) return r;
    uint32_t l = 0;
    while ((1ull << l) < d) ++l; // ceil(log2 d) >= 1
    r.shift = l - 1u;
    r.mul = (uint32_t)((((unsigned long long)1 << (32u + r.shift)) + d - 1u) / d);
    return r;
}

// the view of ONE frame of a launch sequence whose frames have cameras of their own (rptr_hip_render_batch_cameras_async; the reference's
// loop may move the camera every frame: app.cpp:350-469, vulkan/render_vulkan.cpp:2880-2941)
struct RpCam {
    float pos[3], du[3], dv[3], dir_top_left[3];
};
#define RP_BATCH_CAMS 8 // frames of a launch sequence that can have a camera of their own
// the per-frame constants: RenderParams + SceneParams + ViewParams subset
// (vulkan/gpu_params.glsl:61-87,120-131) + this backend's tile mapping
struct RpFrame {
    RptrRenderParams rp;
    RptrSceneParams sp;
    RptrLightSamplingConfig lc;
    float cam_pos[3];
    uint32_t frame_offset;
    float cam_du[3];
    uint32_t sample_base; // sample_index of sample slot 0 of this batch
    float cam_dv[3];
    int32_t batch_spp;    // sample slots in flight
    float cam_dir_top_left[3];
    int32_t variant;
    int32_t width, height;       // full frame
    int32_t local_rows;          // rows owned by this rank
    int32_t tiles_x, tiles_y;    // 8x8 pixel tiles over the local image
    int32_t npix_padded;         // tiles_x*tiles_y*64
    int32_t rank, world, stripe_rows;
    int32_t num_bins;            // SCENE_GET_BINNED_LIGHTS_BIN_COUNT (pt_megakernel.glsl:103)
    int32_t alpha_test;          // the scene has alpha-tested materials: the first extend hands its generator on in the path state
    uint32_t frame_id;           // view_params.frame_id: samples accumulated before this render call (render_vulkan.cpp:2913)
    // AOV images (vulkan/accumulate.glsl:76-103; RGBA16F, local pixels row by row), NULL = off. The first sample of a frame
    // writes them at bounce 0.
    uint2 *aov_albedo_roughness, *aov_normal_depth, *aov_motion_jitter;
    // x / y / w rows of VP and VP_reference (render_vulkan.cpp:2926-2931): world -> view (3x4 row-major) and the two scale
    // factors of the infinite perspective, clip = (P00 v.x, -P11 v.y, ., -v.z)
    float view[12], view_ref[12], proj[2], proj_ref[2];
    int32_t regroup_materials;   // RPTR_REGROUP=1: shade orders the hits of a chunk by material inside its LDS compaction (kernels.h)
    int32_t _pad_regroup;
    RpDivU32 div_npix_padded, div_tiles_x, div_stripe_rows, div_width; // rp_div by npix_padded / tiles_x / stripe_rows / width
    // A launch sequence may carry the samples of SEVERAL frames (rptr_hip_render_batch_async): sample slots [k * frame_spp, (k+1) *
    // frame_spp) belong to frame k of the batch. Frame 0 has (frame_offset, sample_base, frame_id) above; the frames behind it all reset
    // the accumulation (batch_reset != 0: begin_frame's frame_offset += frame_id, frame_id = 0 per frame) or all continue it.
    int32_t batch_frames;        // frames in this launch sequence (1: the whole batch is one frame)
    int32_t frame_spp;           // sample slots per frame
    int32_t batch_reset;         // frames 1.. restart the accumulation
    RpDivU32 div_frame_spp;
    size_t out_stride;           // pixels between the per-frame output images of a batch (out_accum / out_fb of rp_k_resolve)
    // point set (rptr_hip_set_rng_variant): RPTR_RNG_VARIANT_* and its table as the reference uploads it (SobolData: 1024 x 32 matrix
    // words + the 256 x 256 tile inversion; BNData: 256 x 256 sequence values + the 128 x 128 x 8 scrambling tile)
    int32_t rng_variant;
    int32_t per_frame_cams;      // the frames of this launch sequence have their own cameras (cams[frame]); 0: cam_* above serves them all
    const uint32_t *rng_table;
    // cam_* above is frame 0's camera; view / proj / view_ref / proj_ref and aov_cam_pos belong to the LAST frame of the sequence (the one
    // whose first sample writes the AOV images)
    float aov_cam_pos[3];
    float _pad_cam;
    RpCam cams[RP_BATCH_CAMS];
};
// the frame a sample slot belongs to and its frame constants: sample_index, frame_offset (lcg_rng.glsl:36-39) and view_params.frame_id
// (samples accumulated before the frame: seeds the alpha test of shadow rays, pt_megakernel.glsl:251-262)
struct RpSlotFrame {
    uint32_t frame, sample_index, frame_offset, frame_id;
};
#ifdef __HIPCC__
RP_DEV RpSlotFrame rp_slot_frame(const RpFrame &f, uint32_t sslot) {
    RpSlotFrame r;
    r.frame = f.batch_frames > 1 ? rp_div(sslot, f.div_frame_spp) : 0u;
    if (r.frame == 0u) {
        r.sample_index = f.sample_base + sslot;
        r.frame_offset = f.frame_offset;
        r.frame_id = f.frame_id;
    } else if (f.batch_reset) { // begin_frame of every further frame: frame_offset += frame_id (= what was accumulated), frame_id = 0
        r.sample_index = sslot - r.frame * uint32_t(f.frame_spp);
        r.frame_offset = f.frame_offset + f.sample_base + r.frame * uint32_t(f.frame_spp);
        r.frame_id = 0u;
    } else {
        r.sample_index = f.sample_base + sslot;
        r.frame_offset = f.frame_offset;
        r.frame_id = f.frame_id + r.frame * uint32_t(f.frame_spp);
    }
    return r;
}
#endif

// local tiled slot -> local pixel; false for padding lanes. Tiles are numbered in BLOCKS of RP_TILE_BLOCK x RP_TILE_BLOCK (RpFrame::tiles_x / tiles_y are
// multiples of it, div_tiles_x divides by tiles_x / RP_TILE_BLOCK). 2 x 2 (round 5): the four consecutive tiles a traversal wave takes as one pool
// are a 16 x 16 pixel square instead of a 32 x 8 strip -- its rays share more of the tree: closest-hit launches -2...4 % (C2, C3, C5), a
// 1/8 frame's -10 %, one frame at a time 1.86 -> 1.80 ms; 4 x 4 blocks (with or without Z order inside) and 8 x 8: no better than row-major
// (profiles/r05_notes.md section 17). -DRP_TILE_BLOCK=1: row-major tiles, rounds 1-5.
#ifndef RP_TILE_BLOCK
#define RP_TILE_BLOCK 2
#endif
RP_DEV bool rp_slot_to_local(const RpFrame &f, uint32_t slot, int &lx, int &ly) {
    uint32_t tile = slot >> 6, in = slot & 63u;
    const uint32_t blk = tile / uint32_t(RP_TILE_BLOCK * RP_TILE_BLOCK), t_in = tile % uint32_t(RP_TILE_BLOCK * RP_TILE_BLOCK);
    const uint32_t by = rp_div(blk, f.div_tiles_x);
    const uint32_t bx = blk - by * (uint32_t(f.tiles_x) / uint32_t(RP_TILE_BLOCK));
    const uint32_t tix = t_in % uint32_t(RP_TILE_BLOCK), tiy = t_in / uint32_t(RP_TILE_BLOCK);
    const int tx = int(bx * uint32_t(RP_TILE_BLOCK) + tix), ty = int(by * uint32_t(RP_TILE_BLOCK) + tiy);
    lx = tx * 8 + int(in & 7u);
    ly = ty * 8 + int(in >> 3);
    return lx < f.width && ly < f.local_rows;
}
RP_DEV uint32_t rp_local_to_slot(const RpFrame &f, int lx, int ly) {
    const uint32_t tx = uint32_t(lx >> 3), ty = uint32_t(ly >> 3);
    const uint32_t blk = (ty / uint32_t(RP_TILE_BLOCK)) * (uint32_t(f.tiles_x) / uint32_t(RP_TILE_BLOCK)) + tx / uint32_t(RP_TILE_BLOCK);
    const uint32_t t_in = (ty % uint32_t(RP_TILE_BLOCK)) * uint32_t(RP_TILE_BLOCK) + tx % uint32_t(RP_TILE_BLOCK);
    const uint32_t tile = blk * uint32_t(RP_TILE_BLOCK * RP_TILE_BLOCK) + t_in;
    return tile * 64u + uint32_t((ly & 7) * 8 + (lx & 7));
}
// local row -> frame row (stripe s of `stripe_rows` rows belongs to rank s % world)
RP_DEV int rp_local_row_to_global(const RpFrame &f, int ly) {
    int stripe_local = int(rp_div(uint32_t(ly), f.div_stripe_rows));
    return (stripe_local * f.world + f.rank) * f.stripe_rows + (ly - stripe_local * f.stripe_rows);
}

// ------------------------------------------------------------------ AOV stores (vulkan/accumulate.glsl:76-103)
RP_DEV uint2 rp_half4(float a, float b, float c, float d) { // RGBA16F texel, round to nearest even
    const uint32_t ua = __builtin_bit_cast(uint16_t, (_Float16)a), ub = __builtin_bit_cast(uint16_t, (_Float16)b);
    const uint32_t uc = __builtin_bit_cast(uint16_t, (_Float16)c), ud = __builtin_bit_cast(uint16_t, (_Float16)d);
    return make_uint2(ua | (ub << 16), uc | (ud << 16));
}
RP_DEV void rp_project(const float *view, const float *proj, V3 p, float &x, float &y, float &w) {
    const float vx = ((view[0] * p.x + view[1] * p.y) + view[2] * p.z) + view[3];
    const float vy = ((view[4] * p.x + view[5] * p.y) + view[6] * p.z) + view[7];
    const float vz = ((view[8] * p.x + view[9] * p.y) + view[10] * p.z) + view[11];
    x = proj[0] * vx;
    y = -(proj[1] * vy);
    w = -vz;
}
// view_params.screen_jitter (render_vulkan.cpp:2917-2926): with render_params.enable_raster_taa > 0 all primary rays of a frame share one
// sub-pixel offset, entry (frame_offset + frame_id) mod RASTER_TAA_NUM_SAMPLES (16, CMakeLists.txt:30) of the (2, 3) Halton sequence at the
// six decimals librender/halton.h tabulates it with, in clip-space units: h * 2 / dims - 1 / dims. Zero otherwise.
RP_DEV V2 rp_screen_jitter(const RpFrame &f, uint32_t frame_offset, uint32_t frame_id) {
    static constexpr float halton_23[32] = {0.500000f, 0.333333f, 0.250000f, 0.666667f, 0.750000f, 0.111111f, 0.125000f, 0.444444f, 0.625000f, 0.777778f, 0.375000f, 0.222222f, 0.875000f, 0.555556f, 0.062500f, 0.888889f, 0.562500f, 0.037037f, 0.312500f, 0.370370f, 0.812500f, 0.703704f, 0.187500f, 0.148148f, 0.687500f, 0.481481f, 0.437500f, 0.814815f, 0.937500f, 0.259259f, 0.031250f, 0.592593f};
    if (f.rp.enable_raster_taa <= 0) return v2(0.0f, 0.0f);
    const uint32_t idx = (frame_offset + frame_id) & 15u;
    const float w = float(f.width), h = float(f.height);
    return v2(halton_23[2 * idx] * 2.0f / w - 1.0f / w, halton_23[2 * idx + 1] * 2.0f / h - 1.0f / h);
}
RP_DEV void rp_store_geometry_aovs(const RpFrame &f, int pixel, V3 normal, V3 hit_point, V2 screen_jitter) { // :76-96 (motion_vector = 0)
    f.aov_normal_depth[pixel] = rp_half4(normal.x, normal.y, normal.z, len3(hit_point - ld3(f.aov_cam_pos)));
    float rx, ry, rw, cx, cy, cw;
    rp_project(f.view_ref, f.proj_ref, hit_point, rx, ry, rw);
    rp_project(f.view, f.proj, hit_point, cx, cy, cw);
    const float rd = fmaxf(rw, 0.0f), cd = fmaxf(cw, 0.0f);
    f.aov_motion_jitter[pixel] = rp_half4(rx / rd - cx / cd, ry / rd - cy / cd, screen_jitter.x, screen_jitter.y);
}
RP_DEV void rp_store_material_aovs(const RpFrame &f, int pixel, V3 albedo, float roughness, float ior) { // :98-103
    f.aov_albedo_roughness[pixel] = rp_half4(albedo.x, albedo.y, albedo.z, ior != 1.0f ? roughness : 1.0f);
}

// ------------------------------------------------------------------ RNG (a2)
// rendering/pointsets/hashing.glsl:11-39
RP_DEV uint32_t rp_murmur_mix(uint32_t hash, uint32_t k) {
    k *= 0xcc9e2d51u;
    k = (k << 15) | (k >> 17);
    k *= 0x1b873593u;
    hash ^= k;
    hash = ((hash << 13) | (hash >> 19)) * 5u + 0xe6546b64u;
    return hash;
}
RP_DEV uint32_t rp_murmur_finalize(uint32_t hash) {
    hash ^= hash >> 16;
    hash *= 0x85ebca6bu;
    hash ^= hash >> 13;
    hash *= 0xc2b2ae35u;
    hash ^= hash >> 16;
    return hash;
}
// rendering/pointsets/lcg_rng.glsl:28-39
RP_DEV uint32_t rp_rng_seed(uint32_t index, uint32_t frame, uint32_t px, uint32_t py, uint32_t dimx) {
    uint32_t s = rp_murmur_mix(frame, px + py * dimx);
    s = rp_murmur_mix(s, index);
    return rp_murmur_finalize(s);
}
// rendering/pointsets/lcg_rng.glsl:15-26; float(u32) * 2^-32 == ldexp(float(u32), -32), may be exactly 1.0f
RP_DEV float rp_randf(uint32_t &state) {
    state = state * 1664525u + 1013904223u;
    return float(state) * 2.3283064365386962890625e-10f;
}
RP_DEV V2 rp_rand2(uint32_t &state) { // rendering/defaults.glsl:29-35
    V2 r;
    r.x = rp_randf(state);
    r.y = rp_randf(state);
    return r;
}

// ------------------------------------------------------------------ point sets (RBO rng_variant, render_params.glsl.h:34-37)
// The generator of a path: RANDOM_STATE of rendering/pointsets/{lcg_rng,sobol,bn_rng}.glsl behind one type.
//   uniform: `s` is the LCG state, dimensions are ignored (defaults.glsl:23-50).
//   Sobol / Z-Sobol: `index` is the point of the sequence, every draw XORs a fresh LCG number (`s`: the scramble) into it (sobol.glsl:197-206).
//   blue noise: `index` = sampleID, `pix` = pixelID inside the 128 x 128 tile (bn_rng.glsl:80-92); no state changes between draws.
// Only `s` lives in the path state (ray_d.w); index / pix are functions of the path id and are recomputed by rp_rng_open.
struct RpRng {
    uint32_t s, index, pix;
};
#define RP_SOBOL_DIMS 1024u   // sobol_data.h:7-11
#define RP_SOBOL_BITS 32u
#define RP_SOBOL_TILE 256u
#define RP_BN_SAMPLES 256u    // bn_data.h:7-10
#define RP_BN_DIMS 256u
#define RP_BN_SCR_DIMS 8u
#define RP_BN_TILE 128u
RP_DEV uint32_t rp_part1by1(uint32_t x) { // rendering/util.glsl:156-163 (bits of x spread to the even positions)
    x &= 0x0000ffffu;
    x = (x ^ (x << 8)) & 0x00ff00ffu;
    x = (x ^ (x << 4)) & 0x0f0f0f0fu;
    x = (x ^ (x << 2)) & 0x33333333u;
    x = (x ^ (x << 1)) & 0x55555555u;
    return x;
}
// sample_order.glsl:22-73 morton_sample_id(0, pixel, uvec2(tile), hash_tile_id = true, hash_sample_id = false) for a square power-of-two tile:
// the Z-order position of the pixel inside its tile, every bit pair permuted (and possibly swapped) by a hash of the bits above it.
RP_DEV uint32_t rp_morton_shuffled(uint32_t px, uint32_t py, uint32_t tile) {
    const uint32_t pcount = tile * tile, mask1 = tile - 1u, mask2 = pcount - 1u;
    const uint32_t ex = rp_part1by1(px), ey = rp_part1by1(py);
    uint32_t id = ((ey << 1) + ex) & mask2;
    id |= ((px | py) & ~mask1) * tile; // bits present in one dimension only go on top (they take part in the hashes: hash_tile_id)
    uint32_t swap_bits = ex ^ ey;
    swap_bits |= swap_bits << 1;
    uint32_t out = id;
    for (uint32_t ie = 2u * uint32_t(31 - __clz(int(tile))); ie > 0u;) {
        uint32_t perm = rp_murmur_finalize(rp_murmur_mix(0u, id >> ie));
        const bool swap = (perm & 4u) != 0u;
        perm &= 3u;
        ie -= 2u;
        out ^= (perm << ie) & mask2;
        const uint32_t swap_mask = swap ? (3u << ie) : 0u;
        if (swap_mask == (mask2 & swap_mask)) out ^= swap_bits & swap_mask;
    }
    return out & mask2;
}
// sobol.glsl:112-130: the next point after `index_shift` whose first two coordinates fall on the same pixel of the 256 x 256 tile as
// point index_shift + index would without the shift (tile_invert_1_0: pixel -> point, rendering/tools/prepare_sobol.cpp:36-58)
RP_DEV uint32_t rp_sobol_shift_invert(const uint32_t *table, uint32_t index, uint32_t index_shift) {
    index += index_shift;
    uint32_t r0 = 0u, r1 = 0u;
    for (uint32_t i = 0u; index != 0u; index >>= 1, ++i)
        if (index & 1u) {
            r0 ^= table[i];
            r1 ^= table[RP_SOBOL_BITS + i];
        }
    r0 >>= 24;
    r1 >>= 24;
    return index_shift + table[RP_SOBOL_DIMS * RP_SOBOL_BITS + r1 * RP_SOBOL_TILE + r0];
}
// GET_RNG(sample_index, view_params.frame_offset, uvec4(pixel, frame_dims)) of the selected point set (pt_megakernel.glsl:314;
// lcg_rng.glsl:28-39, sobol.glsl:165-195, bn_rng.glsl:80-92 -- whose GET_RNG takes frame_id / frame_offset of the view instead)
// TABLE = false: the uniform generator (RBO default), the table code is compiled out of the kernel
template <bool TABLE>
RP_DEV RpRng rp_rng_open(const RpFrame &f, const RpSlotFrame &sf, uint32_t px, uint32_t py) {
    RpRng r;
    r.index = r.pix = 0u;
    if (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) {
        r.s = rp_rng_seed(sf.sample_index, sf.frame_offset, px, py, uint32_t(f.width));
    } else if (f.rng_variant == RPTR_RNG_VARIANT_BN) {
        r.s = 0u;
        r.pix = (px & (RP_BN_TILE - 1u)) + (py & (RP_BN_TILE - 1u)) * RP_BN_TILE;
        r.index = sf.frame_id + sf.frame_offset * 13u;
    } else {
        uint32_t linear = px + py * uint32_t(f.width); // per-pixel scrambling
        r.index = sf.sample_index;
        if (f.rng_variant == RPTR_RNG_VARIANT_Z_SBL) {
            const uint32_t sample_offset = rp_morton_shuffled(px, py, RP_SOBOL_TILE);
            r.index = rp_sobol_shift_invert(f.rng_table, sample_offset, RP_SOBOL_TILE * RP_SOBOL_TILE * sf.sample_index);
            linear = (px >> 8) + (py >> 8) * (uint32_t(f.width) >> 8); // per-tile scrambling
        }
        // get_lcg_rng(frame_id, 0, linear) with sobol.glsl's "frame_id" = the rnd_offset it is handed
        r.s = rp_murmur_finalize(rp_murmur_mix(rp_murmur_mix(0u, linear), sf.frame_offset));
    }
    return r;
}
// sobol.glsl:74-108 (sobol_point) / bn_rng.glsl:30-71 (sample_bnd, BN_OPTIMIZED_SPP 1: no ranking) for absolute dimension `d`
RP_DEV float rp_draw_table(const RpFrame &f, RpRng &r, uint32_t d) {
    const uint32_t *table = f.rng_table;
    if (f.rng_variant == RPTR_RNG_VARIANT_BN) {
        uint32_t pixel = r.pix, sample = r.index;
        const uint32_t x_doffset = d / RP_BN_SCR_DIMS;
        pixel = ((pixel + x_doffset) & (RP_BN_TILE - 1u)) + (pixel & ~(RP_BN_TILE - 1u));
        d = (d & (RP_BN_SCR_DIMS - 1u)) + x_doffset / RP_BN_TILE * RP_BN_SCR_DIMS;
        d &= RP_BN_DIMS - 1u;
        if (sample & 1u) pixel ^= RP_BN_TILE - 1u;
        if (sample & 2u) pixel ^= (RP_BN_TILE - 1u) * RP_BN_TILE;
        const uint32_t x_soffset = sample * 73u, y_soffset = sample * 97u;
        pixel = ((pixel + x_soffset) & (RP_BN_TILE - 1u)) + (pixel & ~(RP_BN_TILE - 1u));
        pixel = ((pixel + y_soffset * RP_BN_TILE) & (RP_BN_TILE * (RP_BN_TILE - 1u))) + (pixel & ~(RP_BN_TILE * (RP_BN_TILE - 1u)));
        sample = 0u; // sampleID & (BN_OPTIMIZED_SPP - 1)
        const uint32_t ranking_index = pixel * RP_BN_SCR_DIMS + (d & (RP_BN_SCR_DIMS - 1u));
        uint32_t value = table[d + sample * RP_BN_DIMS];
        value ^= table[RP_BN_SAMPLES * RP_BN_DIMS + ranking_index];
        return (0.5f + float(value)) / 256.0f;
    }
    r.s = r.s * 1664525u + 1013904223u; // lcg_random(rng.scramble)
    d &= RP_SOBOL_DIMS - 1u;
    uint32_t result = r.s;
    uint32_t index = r.index;
    for (uint32_t i = d * RP_SOBOL_BITS; index != 0u; index >>= 1, ++i)
        if (index & 1u) result ^= table[i];
    if (f.rng_variant == RPTR_RNG_VARIANT_Z_SBL && d < 2u) result ^= result << 8; // sobol.glsl:88-102
    return float(result) * 2.3283064365386962890625e-10f;
}
// RANDOM_FLOAT1 / RANDOM_FLOAT2(rng, dim) with the dimension already made absolute (RANDOM_SET_DIM / RANDOM_SHIFT_DIM are plain
// additions: the callers pass DIM_CAMERA_END + bounce * (DIM_VERTEX_END + DIM_LIGHT_END) + ..., rendering/pathspace.h)
template <bool TABLE>
RP_DEV float rp_draw1(const RpFrame &f, RpRng &r, uint32_t d) {
    if (!TABLE || f.rng_variant == RPTR_RNG_VARIANT_UNIFORM) return rp_randf(r.s);
    return rp_draw_table(f, r, d);
}
template <bool TABLE>
RP_DEV V2 rp_draw2(const RpFrame &f, RpRng &r, uint32_t d) { // defaults.glsl:29-35: x first
    V2 v;
    v.x = rp_draw1<TABLE>(f, r, d);
    v.y = rp_draw1<TABLE>(f, r, d + 1u);
    return v;
}
#define RP_DIM_CAMERA_END 6u   // pathspace.h:17 (the megakernel does not define USE_SIMPLIFIED_CAMERA)
#define RP_DIM_BOUNCE 8u       // DIM_VERTEX_END + DIM_LIGHT_END
RP_DEV uint32_t rp_bounce_dim(int bounce) { return RP_DIM_CAMERA_END + uint32_t(bounce) * RP_DIM_BOUNCE; }

// ------------------------------------------------------------------ util.glsl
RP_DEV void rp_ortho_basis(V3 &v_x, V3 &v_y, V3 n) { // rendering/util.glsl:73-87
    v_y = v3(0, 0, 0);
    if (n.x < 0.6f && n.x > -0.6f)
        v_y.x = 1.f;
    else if (n.y < 0.6f && n.y > -0.6f)
        v_y.y = 1.f;
    else if (n.z < 0.6f && n.z > -0.6f)
        v_y.z = 1.f;
    else
        v_y.x = 1.f;
    v_x = norm3(cross3(v_y, n));
    v_y = norm3(cross3(n, v_x));
}
RP_DEV float rp_cos_half_angle(float c) { return rp_fdiv_sqrt(1.0f + c, 2.0f + 2.0f * c); } // util.glsl:120-122
RP_DEV float rp_mix_fma(float x, float y, float a) { return fmaf(a, y, fmaf(-a, x, x)); }   // util.glsl:151-153
RP_DEV float rp_linear_to_srgb(float x) {                                                  // util.glsl:19-28
    return (x <= 0.0031308f) ? 12.92f * x : 1.055f * powf(fmaxf(fabsf(x), 1.192092896e-07f), 1.f / 2.4f) - 0.055f;
}

// ------------------------------------------------------------------ dequantisation (a6, a7)
RP_DEV V3 rp_dequantize_position(uint64_t w, V3 scaling, V3 offset) { // librender/dequantize.glsl:8-21
    V3 q = v3(float(uint32_t(w) & 0x1FFFFFu), float(uint32_t(w >> 21) & 0x1FFFFFu), float(uint32_t(w >> 42) & 0x1FFFFFu));
    return q * scaling + offset;
}
RP_DEV V3 rp_dequantize_normal(uint32_t word) { // librender/dequantize.glsl:23-41
    V2 n = v2(float(int(word & 0xFFFFu) - 0x8000), float(int(word >> 16) - 0x8000)) / float(0x7FFF);
    float nl1 = fabsf(n.x) + fabsf(n.y);
    if (nl1 >= 1.0f)
        n = (v2(1.0f, 1.0f) - v2(fabsf(n.y), fabsf(n.x))) * v2(n.x >= 0.0f ? 1.0f : -1.0f, n.y >= 0.0f ? 1.0f : -1.0f);
    return norm3(v3(n.x, n.y, 1.0f - nl1));
}
RP_DEV V2 rp_dequantize_uv(uint32_t word) { // librender/dequantize.glsl:43-48
    return v2(0.0f, 1.0f) + v2(float(int(word & 0xFFFFu)), float(-int(word >> 16))) * (8.0f / float(0xFFFFu));
}

// ------------------------------------------------------------------ hit attributes (a7)
struct RpHit { // rendering/rt/hit.glsl:12-23
    V3 normal;
    float dist;
    V3 geo_normal;
    int material_id;
    V3 tangent;
    float bitangent_l;
    V2 uv;
};
RP_DEV int rp_hit_material_id(const RpGeomRecord &g, uint32_t prim) { // rendering/rt/hit.glsl:49-56
    if (g.material_id < 0)
        return int(g.mat_ids[prim]) - g.material_id - 1;
    return g.material_id;
}
RP_DEV void rp_geom_tri(const RpGeomRecord &g, uint32_t prim, V3 &a, V3 &b, V3 &c) {
    if (g.flags & RP_GEOM_DYNAMIC) { // pt_megakernel.glsl:526-529
        const float *p = g.dyn_pos + 9ull * prim;
        a = v3(p[0], p[1], p[2]);
        b = v3(p[3], p[4], p[5]);
        c = v3(p[6], p[7], p[8]);
        return;
    }
    V3 sc = ld3(g.scaling), of = ld3(g.offset);
    const uint64_t *q = g.qpos + 3ull * prim;
    a = rp_dequantize_position(q[0], sc, of);
    b = rp_dequantize_position(q[1], sc, of);
    c = rp_dequantize_position(q[2], sc, of);
}
// the uv part of calc_hit_attributes alone (hit.glsl:99-101): what the alpha test of a candidate needs
RP_DEV V2 rp_hit_uv(const RpGeomRecord &g, uint32_t prim, float bu, float bv) {
    if ((g.flags & RP_GEOM_HAS_UVS) == 0) return v2(0.0f, 0.0f);
    const V3 bary = v3(1.f - bu - bv, bu, bv);
    const uint64_t *q = g.qnrm_uv + 3ull * prim;
    const V2 uva = rp_dequantize_uv(uint32_t(q[0] >> 32)), uvb = rp_dequantize_uv(uint32_t(q[1] >> 32)), uvc = rp_dequantize_uv(uint32_t(q[2] >> 32));
    return v2((uva.x * bary.x + uvb.x * bary.y) + uvc.x * bary.z, (uva.y * bary.x + uvb.y * bary.y) + uvc.y * bary.z);
}
// rendering/rt/hit.glsl:58-128 via the quantised overload :162-203; the vertices va / vb / vc and the three qnrm_uv words come from the
// triangle's shading record (RpShadeTri: the values rp_geom_tri and the qnrm_uv stream hold)
RP_DEV RpHit rp_calc_hit_attributes(V3 va, V3 vb, V3 vc, uint64_t qa, uint64_t qb, uint64_t qc, bool has_normals, bool has_uvs, int material_id, float ray_t,
                                    float bu, float bv, const M3 &normals_to_world) {
    RpHit h;
    h.dist = ray_t;
    V3 gn = cross3(vb - va, vc - va);
    V3 n = gn;
    const V3 bary = v3(1.f - bu - bv, bu, bv);
    if (has_normals) {
        M3 nm{rp_dequantize_normal(uint32_t(qa)), rp_dequantize_normal(uint32_t(qb)), rp_dequantize_normal(uint32_t(qc))};
        n = mul(nm, bary);
        if (dot3(n, gn) < 0.0f) gn = -gn;
    }
    h.geo_normal = gn * 0.5f;
    h.normal = n;
    h.material_id = material_id;
    bool requires_tangent = true;
    h.geo_normal = mul(normals_to_world, h.geo_normal);
    h.normal = norm3(mul(normals_to_world, h.normal));
    h.uv = v2(0.0f, 0.0f);
    if (has_uvs) {
        V2 uva = rp_dequantize_uv(uint32_t(qa >> 32)), uvb = rp_dequantize_uv(uint32_t(qb >> 32)), uvc = rp_dequantize_uv(uint32_t(qc >> 32));
        h.uv = v2((uva.x * bary.x + uvb.x * bary.y) + uvc.x * bary.z, (uva.y * bary.x + uvb.y * bary.y) + uvc.y * bary.z); // uvs * bary
        float posframe_det = len3(gn);
        V3 frame_n = gn / (posframe_det * posframe_det);
        V3 dp2perp = cross3(vc - va, frame_n);
        V3 dp1perp = cross3(frame_n, vb - va);
        V2 duv1 = uvb - uva, duv2 = uvc - uva;
        V3 T = dp2perp * duv1.x + dp1perp * duv2.x;
        V3 B = dp2perp * duv1.y + dp1perp * duv2.y;
        T = mul(normals_to_world, T);
        B = mul(normals_to_world, B);
        float Tlen = len3(T);
        if (Tlen > 0.0f && !isinf(Tlen) && !isnan(Tlen)) {
            h.tangent = T;
            h.bitangent_l = dot3(norm3(cross3(h.geo_normal, T)), B);
            requires_tangent = false;
        }
    }
    if (requires_tangent) {
        h.tangent = norm3(mul(normals_to_world, cross3(vc - va, gn)));
        h.bitangent_l = 1.0f;
    }
    return h;
}
// the shading record of one BVH triangle from the scene tables (what rp_k_build_shade_tris stores)
RP_DEV RpShadeTri rp_make_shade_tri(const RpGeomRecord &g, uint32_t prim) {
    RpShadeTri r;
    V3 a, b, c;
    rp_geom_tri(g, prim, a, b, c);
    r.pos[0] = a.x, r.pos[1] = a.y, r.pos[2] = a.z;
    r.pos[3] = b.x, r.pos[4] = b.y, r.pos[5] = b.z;
    r.pos[6] = c.x, r.pos[7] = c.y, r.pos[8] = c.z;
    const bool has_normals = (g.flags & RP_GEOM_HAS_NORMALS) != 0, has_uvs = (g.flags & RP_GEOM_HAS_UVS) != 0;
    uint64_t qa = 0, qb = 0, qc = 0;
    if (has_normals || has_uvs) {
        const uint64_t *q = g.qnrm_uv + 3ull * prim;
        qa = q[0];
        qb = q[1];
        qc = q[2];
    }
    r.qnu[0] = uint32_t(qa), r.qnu[1] = uint32_t(qa >> 32);
    r.qnu[2] = uint32_t(qb), r.qnu[3] = uint32_t(qb >> 32);
    r.qnu[4] = uint32_t(qc), r.qnu[5] = uint32_t(qc >> 32);
    r.material = (uint32_t(rp_hit_material_id(g, prim)) & RP_SHADE_MATERIAL_MASK) | (has_normals ? RP_SHADE_HAS_NORMALS : 0u) | (has_uvs ? RP_SHADE_HAS_UVS : 0u);
    return r;
}

// ------------------------------------------------------------------ materials (a9)
struct RpMaterial { // GLTFMaterial (gltf_bsdf.glsl:15-35) / SimpleMaterial (simple_bsdf.glsl:18-28)
    V3 base_color;
    float metallic, specular, roughness, ior;
    uint32_t flags;
    // RPTR_VARIANT_GLTF_TRANSMISSION only (GLTF_SUPPORT_TRANSMISSION[_ROUGHNESS])
    float transmission_roughness, specular_transmission;
    V3 transmission_color;
};
// ---- texture sampling: the reference's material sampler (render_vulkan.cpp:1657-1670: linear filter, linear mip filter, REPEAT, LOD 0..16,
// anisotropy 12) in software after the Vulkan specification's texel-filtering equations (oracle/oshade.h states them): texel centres at
// (i + 0.5) / size, bilinear weights in float, unorm byte / 255, sRGB decode per texel through the host-computed table before filtering.
//   rp_texture_lod(uv, lod): the two nearest levels blended by the fraction of the clamped lod
//   rp_texture_grad(uv, ddx, ddy): eta = min(rho_max / rho_min, 12), N = ceil(eta) taps along the larger derivative at log2(rho_max / eta)
// Levels are stored back to back, level l = max(1, w >> l) x max(1, h >> l) (RptrTextureDesc.mip_levels).
struct RpTexCoord { // HitPoint::uv + HitPoint::duvdxy
    V2 uv, ddx, ddy;
};
RP_DEV RpTexCoord rp_texcoord(V2 uv) { return RpTexCoord{uv, v2(0.0f, 0.0f), v2(0.0f, 0.0f)}; }
struct RpMipView {
    const uchar4 *texels;
    int w, h;
};
RP_DEV RpMipView rp_mip_view(const RpTexture &t, int level) {
    RpMipView v{t.texels, t.width, t.height};
    for (int l = 0; l < level; ++l) {
        v.texels += (size_t)v.w * (size_t)v.h;
        if (v.w > 1) v.w /= 2;
        if (v.h > 1) v.h /= 2;
    }
    return v;
}
RP_DEV float4 rp_texel(const RpScene &sc, const RpTexture &t, const RpMipView &v, int ix, int iy) {
    const uchar4 c = v.texels[(size_t)iy * (size_t)v.w + (size_t)ix];
    if (t.srgb) return make_float4(sc.srgb_lut[c.x], sc.srgb_lut[c.y], sc.srgb_lut[c.z], rp_fdiv(float(c.w), 255.0f));
    return make_float4(rp_fdiv(float(c.x), 255.0f), rp_fdiv(float(c.y), 255.0f), rp_fdiv(float(c.z), 255.0f), rp_fdiv(float(c.w), 255.0f));
}
RP_DEV int rp_wrap_repeat(int i, int n) {
    if ((n & (n - 1)) == 0) return i & (n - 1); // power-of-two sizes (the usual case): no integer division (~30 instructions each, 4 per tap)
    i %= n;
    return i < 0 ? i + n : i;
}
RP_DEV float4 rp_texture_bilinear(const RpScene &sc, const RpTexture &t, const RpMipView &mv, V2 uv) {
    const float x = uv.x * float(mv.w) - 0.5f, y = uv.y * float(mv.h) - 0.5f;
    const float x0 = floorf(x), y0 = floorf(y);
    const float fx = x - x0, fy = y - y0;
    const int ix0 = rp_wrap_repeat(int(x0), mv.w), ix1 = rp_wrap_repeat(int(x0) + 1, mv.w);
    const int iy0 = rp_wrap_repeat(int(y0), mv.h), iy1 = rp_wrap_repeat(int(y0) + 1, mv.h);
    const float4 c00 = rp_texel(sc, t, mv, ix0, iy0), c10 = rp_texel(sc, t, mv, ix1, iy0), c01 = rp_texel(sc, t, mv, ix0, iy1), c11 = rp_texel(sc, t, mv, ix1, iy1);
    const float gx = 1.0f - fx, gy = 1.0f - fy;
    const float4 top = make_float4(c00.x * gx + c10.x * fx, c00.y * gx + c10.y * fx, c00.z * gx + c10.z * fx, c00.w * gx + c10.w * fx);
    const float4 bot = make_float4(c01.x * gx + c11.x * fx, c01.y * gx + c11.y * fx, c01.z * gx + c11.z * fx, c01.w * gx + c11.w * fx);
    return make_float4(top.x * gy + bot.x * fy, top.y * gy + bot.y * fy, top.z * gy + bot.z * fy, top.w * gy + bot.w * fy);
}
RP_DEV float4 rp_texture_bilinear(const RpScene &sc, const RpTexture &t, int level, V2 uv) { return rp_texture_bilinear(sc, t, rp_mip_view(t, level), uv); }
RP_DEV float4 rp_texture_lod0(const RpScene &sc, int tex_id, V2 uv) { return rp_texture_bilinear(sc, sc.textures[tex_id], 0, uv); }
// the two levels a clamped lod blends and the weight of the coarser one (0: the finer level alone), chosen once for all taps of a lookup
struct RpLodPick {
    RpMipView fine, coarse;
    float delta;
};
RP_DEV RpLodPick rp_pick_levels(const RpTexture &t, float lod) {
    lod = fminf(fmaxf(lod, 0.0f), fminf(float(t.levels - 1), 16.0f)); // (a NaN lod ends up at level 0)
    RpLodPick k;
    k.delta = 0.0f;
    if (!(lod > 0.0f)) {
        k.fine = k.coarse = rp_mip_view(t, 0);
        return k;
    }
    const float hi = floorf(lod);
    k.delta = lod - hi;
    k.fine = rp_mip_view(t, int(hi));
    k.coarse = k.fine;
    if (k.delta != 0.0f) { // one level further
        k.coarse.texels += (size_t)k.fine.w * (size_t)k.fine.h;
        if (k.coarse.w > 1) k.coarse.w /= 2;
        if (k.coarse.h > 1) k.coarse.h /= 2;
    }
    return k;
}
RP_DEV float4 rp_texture_tap(const RpScene &sc, const RpTexture &t, const RpLodPick &k, V2 uv) {
    const float4 a = rp_texture_bilinear(sc, t, k.fine, uv);
    if (k.delta == 0.0f) return a;
    const float4 b = rp_texture_bilinear(sc, t, k.coarse, uv);
    const float g = 1.0f - k.delta;
    return make_float4(a.x * g + b.x * k.delta, a.y * g + b.y * k.delta, a.z * g + b.z * k.delta, a.w * g + b.w * k.delta);
}
RP_DEV float4 rp_texture_lod(const RpScene &sc, int tex_id, V2 uv, float lod) {
    const RpTexture t = sc.textures[tex_id];
    return rp_texture_tap(sc, t, rp_pick_levels(t, lod), uv);
}
#define RP_MAX_ANISOTROPY 12.0f
RP_DEV float4 rp_texture_grad(const RpScene &sc, int tex_id, const RpTexCoord &tc) {
    const RpTexture t = sc.textures[tex_id];
    const float w = float(t.width), h = float(t.height);
    const float mxx = tc.ddx.x * w, mxy = tc.ddx.y * h, myx = tc.ddy.x * w, myy = tc.ddy.y * h;
    const float rx = rp_fsqrt(mxx * mxx + mxy * mxy), ry = rp_fsqrt(myx * myx + myy * myy);
    const float rmax = fmaxf(rx, ry), rmin = fminf(rx, ry);
    // magnification (the footprint lies inside one texel), or nothing to filter (a 1 x 1 texture): one bilinear tap of level 0
    if (!(rmax > 1.0f) || (t.width == 1 && t.height == 1)) return rp_texture_bilinear(sc, t, 0, tc.uv);
    const float eta = rmin > 0.0f ? fminf(rp_fdiv(rmax, rmin), RP_MAX_ANISOTROPY) : RP_MAX_ANISOTROPY;
    const int n = int(ceilf(eta));
    const RpLodPick pick = rp_pick_levels(t, log2f(rp_fdiv(rmax, eta)));
    const V2 major = rx > ry ? tc.ddx : tc.ddy;
    float4 sum = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = 1; i <= n; ++i) {
        const float at = rp_fdiv(float(i), float(n + 1)) - 0.5f;
        const float4 c = rp_texture_tap(sc, t, pick, v2(tc.uv.x + major.x * at, tc.uv.y + major.y * at));
        sum = make_float4(sum.x + c.x, sum.y + c.y, sum.z + c.z, sum.w + c.w);
    }
    const float inv = rp_frcp(float(n));
    return make_float4(sum.x * inv, sum.y * inv, sum.z * inv, sum.w * inv);
}
// ---- the pixel footprint a path carries for its texture lookups (rendering/rt/footprint.glsl; column-major 2 x 2, same operation order
// as oracle/oshade.h)
struct M2 {
    V2 c0, c1;
};
RP_DEV V2 mul2(const M2 &m, V2 v) { return v2(m.c0.x * v.x + m.c1.x * v.y, m.c0.y * v.x + m.c1.y * v.y); }
RP_DEV M2 mul2(const M2 &a, const M2 &b) { return M2{mul2(a, b.c0), mul2(a, b.c1)}; }
RP_DEV M2 transpose2(const M2 &m) { return M2{v2(m.c0.x, m.c1.x), v2(m.c0.y, m.c1.y)}; }
RP_DEV V2 norm2(V2 v) { return v * rp_frsq(v.x * v.x + v.y * v.y); }
RP_DEV M2 rp_dpdxy_to_footprint(V3 ray_dir, V3 dpdx, V3 dpdy) { // :10-15
    V3 t, b;
    rp_ortho_basis(t, b, ray_dir);
    const M2 F{v2(dot3(t, dpdx), dot3(b, dpdx)), v2(dot3(t, dpdy), dot3(b, dpdy))};
    return mul2(F, transpose2(F));
}
RP_DEV M2 rp_transform_footprint(V3 dst_ray_dir, const M3 &T, V3 src_ray_dir, const M2 &F) { // :28-35
    V3 t, b;
    rp_ortho_basis(t, b, src_ray_dir);
    const V3 Tt = mul(T, t), Tb = mul(T, b);
    rp_ortho_basis(t, b, dst_ray_dir);
    const M2 T3{v2(dot3(t, Tt), dot3(b, Tt)), v2(dot3(t, Tb), dot3(b, Tb))};
    return mul2(mul2(T3, F), transpose2(T3));
}
RP_DEV M2 rp_reflect_footprint(V3 dst_ray_dir, V3 src_ray_dir, const M2 &F) { // :38-43
    const V3 n = norm3(dst_ray_dir - src_ray_dir);
    const M3 R{v3(1.0f - 2.0f * (n.x * n.x), 0.0f - 2.0f * (n.y * n.x), 0.0f - 2.0f * (n.z * n.x)),
               v3(0.0f - 2.0f * (n.x * n.y), 1.0f - 2.0f * (n.y * n.y), 0.0f - 2.0f * (n.z * n.y)),
               v3(0.0f - 2.0f * (n.x * n.z), 0.0f - 2.0f * (n.y * n.z), 1.0f - 2.0f * (n.z * n.z))};
    return rp_transform_footprint(dst_ray_dir, R, src_ray_dir, F);
}
RP_DEV void rp_footprint_to_dpdxy(V3 &dpdx, V3 &dpdy, V3 ray_dir, const M2 &F) { // :45-63
    const float B = F.c0.x + F.c1.y;
    const float C = F.c0.x * F.c1.y - F.c0.y * F.c1.x;
    const float D = rp_fsqrt(B * B * 0.25f - C);
    const V2 ev = v2(0.5f * B - D, 0.5f * B + D);
    M2 X;
    if (fabsf(F.c0.y) > 3.0e-39f) {
        X.c0 = v2(F.c1.x, ev.x - F.c0.x);
        X.c1 = v2(ev.y - F.c1.y, F.c0.y);
    } else
        X = M2{v2(1.0f, 0.0f), v2(0.0f, 1.0f)};
    V3 t, b;
    rp_ortho_basis(t, b, ray_dir);
    const V2 x0 = norm2(X.c0) * rp_fsqrt(ev.x), x1 = norm2(X.c1) * rp_fsqrt(ev.y);
    dpdx = t * x0.x + b * x0.y;
    dpdy = t * x1.x + b * x1.y;
}
// the footprint on the surface as uv derivatives (pt_megakernel.glsl:582-606)
RP_DEV RpTexCoord rp_hit_texcoord(V2 uv, const M2 &footprint, V3 ray_dir, V3 geo_normal, V3 tangent, float bitangent_l, float total_t) {
    V3 dpdx, dpdy;
    rp_footprint_to_dpdxy(dpdx, dpdy, ray_dir, footprint);
    const V3 dir_tangent_un = ray_dir - geo_normal * dot3(ray_dir, geo_normal);
    const float cosTheta2 = fmaxf(1.0f - dot3(dir_tangent_un, dir_tangent_un), 0.0f);
    const V3 dir_tangent_elong = dir_tangent_un / (rp_fsqrt(cosTheta2) + cosTheta2);
    const V3 dpdx_ = dpdx + dir_tangent_elong * dot3(dpdx, dir_tangent_un);
    const V3 dpdy_ = dpdy + dir_tangent_elong * dot3(dpdy, dir_tangent_un);
    const V3 bitangent = bitangent_l * cross3(geo_normal, norm3(tangent));
    RpTexCoord tc;
    tc.uv = uv;
    tc.ddx = v2(dot3(tangent, dpdx_), dot3(bitangent, dpdx_)) * total_t;
    tc.ddy = v2(dot3(tangent, dpdy_), dot3(bitangent, dpdy_)) * total_t;
    return tc;
}
// rendering/rt/material_textures.glsl:37-60 (textureGrad: USE_MIPMAPPING, librender/render_params.glsl.h:8)
RP_DEV bool rp_is_textured(float x) { return (__float_as_uint(x) & RPTR_TEXTURED_PARAM_MASK) != 0u; }
// the parameters of a material often read channels of ONE texture at one place (.vks: specular / roughness / metallic): the filtered
// texel of the last lookup is kept, a repeated lookup of the same texture costs nothing (same value: same arithmetic)
struct RpTexelCache {
    int id;
    float4 texel;
};
RP_DEV float4 rp_texture_grad_cached(const RpScene &sc, RpTexelCache &cache, int tex_id, const RpTexCoord &uv) {
    if (cache.id != tex_id) {
        cache.texel = rp_texture_grad(sc, tex_id, uv);
        cache.id = tex_id;
    }
    return cache.texel;
}
RP_DEV float4 rp_textured_color_param(const RpScene &sc, RpTexelCache &cache, float4 x, const RpTexCoord &uv) {
    const uint32_t mask = __float_as_uint(x.x);
    if (mask & RPTR_TEXTURED_PARAM_MASK) return rp_texture_grad_cached(sc, cache, int(RPTR_TEXTURE_ID(mask)), uv);
    return x;
}
RP_DEV float rp_textured_scalar_param(const RpScene &sc, RpTexelCache &cache, float x, const RpTexCoord &uv) {
    const uint32_t mask = __float_as_uint(x);
    if (mask & RPTR_TEXTURED_PARAM_MASK) {
        const float4 t = rp_texture_grad_cached(sc, cache, int(RPTR_TEXTURE_ID(mask)), uv);
        const uint32_t ch = RPTR_TEXTURE_CHANNEL(mask);
        return ch == 0 ? t.x : ch == 1 ? t.y : ch == 2 ? t.z : t.w;
    }
    return x;
}
// The any-hit test of a candidate (pt_megakernel.glsl:153-212 generate_candidate_hit + material_textures.glsl:137-145
// get_material_alpha = the alpha of the base colour parameter at the hit's uv, 1 for a literal colour): true = the
// candidate is ignored. A fractional alpha draws one number from `rng` -- the path's own generator for closest-hit
// queries (RNG_VARIANT_UNIFORM: `#define alpha_rng rng`, :354-358), a generator seeded per candidate for shadow rays.
RP_DEV bool rp_alpha_rejects(const RpScene &sc, int inst_idx, int geom, int prim, float bu, float bv, uint32_t &rng) {
    const int geometry_base = reinterpret_cast<const int *>(sc.insts + inst_idx)[13]; // RptrBvhInstance::geometry_base
    const RpGeomRecord &g = sc.geoms[geometry_base + geom];
    const RptrBaseMaterial &m = sc.materials[rp_hit_material_id(g, uint32_t(prim))];
    if ((m.flags & RPTR_BASE_MATERIAL_NOALPHA) != 0u) return false;
    const uint32_t mask = __float_as_uint(m.base_color[0]);
    float alpha = 1.0f;
    if (mask & RPTR_TEXTURED_PARAM_MASK) alpha = rp_texture_lod0(sc, int(RPTR_TEXTURE_ID(mask)), rp_hit_uv(g, uint32_t(prim), bu, bv)).w;
    if (!(alpha > 0.0f)) return true;
    return alpha < 1.0f && rp_randf(rng) > alpha;
}
// rendering/rt/material_textures.glsl:95-135 (non-unrolled standard textures: a parameter is a literal or a texture handle;
// PREMULTIPLIED_BASE_COLOR_ALPHA is defined, vulkan/gpu_params.glsl:12)
template <int VARIANT, bool TEX>
RP_DEV void rp_unpack_material(const RpScene &sc, RpMaterial &m, V3 &emitter_radiance, const RptrBaseMaterial &p, const RpTexCoord &uv) {
    RpTexelCache cache{-1, make_float4(0.0f, 0.0f, 0.0f, 0.0f)};
    const float4 literal = make_float4(p.base_color[0], p.base_color[1], p.base_color[2], 1.0f);
    const float4 texel = TEX ? rp_textured_color_param(sc, cache, literal, uv) : literal;
    const float alpha = texel.w;
    m.base_color = v3(texel.x, texel.y, texel.z);
    if (alpha > 0.001f) m.base_color = m.base_color / alpha;
    if (VARIANT == RPTR_VARIANT_SIMPLE) { // simple_bsdf.glsl:31-39
        m.roughness = 1.0f;
        m.ior = 1.0f;
        m.metallic = 0.0f;
        m.specular = 0.0f;
    } else {
        m.specular = TEX ? rp_textured_scalar_param(sc, cache, p.specular, uv) : p.specular;
        m.roughness = TEX ? rp_textured_scalar_param(sc, cache, p.roughness, uv) : p.roughness;
        m.metallic = TEX ? rp_textured_scalar_param(sc, cache, p.metallic, uv) : p.metallic;
        m.ior = TEX ? rp_textured_scalar_param(sc, cache, p.ior, uv) : p.ior;
    }
    emitter_radiance = v3(p.base_color[0], p.base_color[1], p.base_color[2]) * p.emission_intensity;
    if (p.emission_intensity != 0.0f) {
        if (TEX && rp_is_textured(p.base_color[0])) emitter_radiance = m.base_color * p.emission_intensity;
        m.base_color = v3s(0.0f);
    }
    if (VARIANT == RPTR_VARIANT_GLTF_TRANSMISSION) { // load_material, gltf_bsdf.glsl:38-62
        m.transmission_roughness = 0.0f;
        m.transmission_color = v3s(0.0f);
        m.specular_transmission = TEX ? rp_textured_scalar_param(sc, cache, p.specular_transmission, uv) : p.specular_transmission;
        if (m.specular_transmission > 0.0f) {
            if (!(m.ior > 1.0f))
                m.specular_transmission = 0.0f; // (the reference folds it into the alpha it returns, which nothing reads any more)
            else {
                m.transmission_color = m.base_color;
                m.transmission_roughness = m.roughness;
                m.roughness = rp_fsqrt(TEX ? rp_textured_scalar_param(sc, cache, p.clearcoat_gloss, uv) : p.clearcoat_gloss);
            }
        }
    }
    m.flags = p.flags;
}

// ------------------------------------------------------------------ glTF BSDF (a14), rendering/bsdfs/gltf_bsdf.glsl
RP_DEV float rp_schlick_weight(float c) { return pow5f(clamp1(1.f - c, 0.f, 1.f)); }                     // :172-174
RP_DEV float rp_gtr_2(float cos_theta_h, float alpha) {                                                  // :194-198
    float a2 = alpha * alpha;
    return rp_fdiv(RP_1_PI * a2, pow2f(1.f + (a2 - 1.f) * cos_theta_h * cos_theta_h));
}
RP_DEV float rp_smith_den1(float n_dot_o, float a2) { return fabsf(n_dot_o) + rp_fsqrt(a2 + (1.0f - a2) * n_dot_o * n_dot_o); } // :200-202
RP_DEV float rp_smith_ggx(float n_dot_o, float n_dot_i, float alpha_g) {                                 // :207-212
    float a = alpha_g * alpha_g;
    float den_shad = rp_smith_den1(n_dot_i, a);
    float den_mask = rp_smith_den1(n_dot_o, a);
    return rp_frcp(den_shad * den_mask);
}
RP_DEV V3 rp_to_pipe_sample(V2 U) { // :216-222
    float phi = 2.0f * RP_PI * U.x;
    return v3(cosf(phi), sinf(phi), U.y);
}
RP_DEV V3 rp_sample_sphere(V3 UP) { // :225-229
    float cos_theta = UP.z * 2.0f - 1.0f;
    float sin_theta = rp_fsqrt(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    return v3(sin_theta * UP.x, sin_theta * UP.y, cos_theta);
}
RP_DEV V3 rp_sample_gtr_2_vndf(V3 w_o_local, float alpha, V3 UP) { // :233-250
    V3 wiStd = norm3(v3(alpha * w_o_local.x, alpha * w_o_local.y, w_o_local.z));
    float z = fmaf((1.0f - UP.z), (1.0f + wiStd.z), -wiStd.z);
    float sinTheta = rp_fsqrt(clamp1(1.0f - z * z, 0.0f, 1.0f));
    float x = sinTheta * UP.x;
    float y = sinTheta * UP.y;
    V3 wmStd = v3(x, y, z) + wiStd;
    V3 wm = v3(wmStd.x * alpha, wmStd.y * alpha, fmaxf(0.0f, wmStd.z));
    float wmL = len3(wm);
    return wm / wmL;
}
RP_DEV float rp_gtr_2_vndf_pdf(float n_dot_o, float cos_theta_h, float alpha) { // :253-257
    return rp_gtr_2(cos_theta_h, alpha) * rp_fdiv(0.5f, rp_smith_den1(n_dot_o, alpha * alpha));
}
RP_DEV V3 rp_gltf_diffuse_basecolor(const RpMaterial &m) { return (1.0f - m.metallic) * m.base_color; } // :259-261
RP_DEV V3 rp_gltf_specular_basecolor(const RpMaterial &m, float ior) {                                  // :263-273
    V3 dielectric_base = v3s(pow2f(rp_fdiv(ior - 1.0f, ior + 1.0f)));
    return mix3(dielectric_base, m.base_color, m.metallic);
}
RP_DEV float rp_gltf_specular_alpha(const RpMaterial &m) { return fmaxf(m.roughness * m.roughness, 0.002f); } // :275-277
RP_DEV float rp_gltf_schlick_weight(float local_o_dot_h, float ior) {                                       // :284-292
    float f_weight = rp_schlick_weight(local_o_dot_h);
    if (ior < 1.0f) {
        float cos_critical = rp_fsqrt(1.0f - ior * ior);
        f_weight = mixf(f_weight, 1.0f, fminf(rp_fdiv(1.0f - local_o_dot_h, 1.0f - cos_critical), 1.0f));
    }
    return f_weight;
}
RP_DEV V3 rp_gltf_bsdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) { // :294-359
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    float ior = o_dot_n < 0.0f ? rp_frcp(m.ior) : m.ior;
    if (i_dot_n * o_dot_n < 0.0f) return v3s(0.0f);
    V3 w_h = norm3(w_i + w_o);
    float o_dot_h = dot3(w_o, w_h);
    V3 diffuse = rp_gltf_diffuse_basecolor(m) * RP_1_PI;
    V3 specular = v3s(0.0f);
    if (m.ior > 1.0f) {
        V3 f0 = rp_gltf_specular_basecolor(m, m.ior);
        float specular_alpha = rp_gltf_specular_alpha(m);
        float specular_refl = rp_gtr_2(dot3(n, w_h), specular_alpha);
        specular_refl *= rp_smith_ggx(o_dot_n, i_dot_n, specular_alpha);
        float f_weight = rp_gltf_schlick_weight(fabsf(o_dot_h), ior);
        V3 F = mix3(f0, v3s(1.0f), f_weight);
        diffuse = diffuse * (v3s(1.0f) - F);
        specular = specular_refl * F;
    }
    return diffuse + specular;
}
struct RpLobes {
    float w0, w1;
};
RP_DEV RpLobes rp_gltf_component_sampler(const RpMaterial &m, float o_dot_h_x, float o_dot_h_y, float vis_x, float vis_y) { // :366-394
    RpLobes c;
    float specular_base_lum = luminance3(rp_gltf_specular_basecolor(m, m.ior));
    float F0 = mixf(specular_base_lum, 1.0f, rp_gltf_schlick_weight(o_dot_h_x, 1.0f));
    float F1 = mixf(specular_base_lum, 1.0f, rp_gltf_schlick_weight(o_dot_h_y, 1.0f));
    c.w0 = (1.0f - F0) * vis_x * (1.0f - m.metallic) * luminance3(rp_gltf_diffuse_basecolor(m));
    c.w1 = F1 * vis_y;
    float weight_sum = 0.0f;
    weight_sum += c.w0;
    weight_sum += c.w1;
    if (weight_sum > 0.0f) {
        c.w0 = rp_fdiv(c.w0, weight_sum);
        c.w1 = rp_fdiv(c.w1, weight_sum);
    } else
        c.w0 = 1.0f;
    return c;
}
RP_DEV float rp_gltf_wpdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) { // :414-494
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    float pdf = RP_1_PI * fabsf(i_dot_n);
    if (m.ior > 1.0f) {
        if (i_dot_n * o_dot_n < 0.0f) return 0.0f;
        V3 w_h = norm3(w_i + w_o);
        float o_dot_h = dot3(w_o, w_h);
        float cos_theta_h = dot3(w_h, n);
        float specular_alpha = rp_gltf_specular_alpha(m);
        float vis_y = rp_fdiv(2.0f * fabsf(i_dot_n), rp_smith_den1(i_dot_n, specular_alpha * specular_alpha));
        RpLobes c = rp_gltf_component_sampler(m, fabsf(o_dot_h), fabsf(o_dot_h), 1.0f, vis_y);
        float specular = rp_gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        pdf *= c.w0;
        pdf += specular * c.w1;
    }
    return pdf;
}
// :496-645. Returns f*|cos|/pdf; pdf == 0 signals "no sample".
RP_DEV V3 rp_sample_gltf_brdf(const RpMaterial &m, V3 n, V3 w_o, V3 &w_i, float &pdf, float &mis_wpdf, V2 rng_sample, V2 fresnel_sample,
                              V3 v_x, V3 v_y) {
    M3 frame{v_x, v_y, n};
    V3 w_o_local = mul_t(frame, w_o);
    float o_dot_n = w_o_local.z;
    mis_wpdf = 0.0f;
    if (o_dot_n < 0.0f) {
        pdf = 0.0f;
        return v3s(0.0f);
    }
    V3 UP = rp_to_pipe_sample(rng_sample);
    V3 w_i_diffuse = norm3(n + rp_sample_sphere(UP));
    float specular_alpha = rp_gltf_specular_alpha(m);
    int component = 0;
    RpLobes lobes{0.0f, 0.0f};
    V3 w_h_specular_local = v3s(0.0f);
    if (m.ior > 1.0f) {
        float odh_x = rp_cos_half_angle(dot3(w_o, w_i_diffuse));
        w_h_specular_local = rp_sample_gtr_2_vndf(w_o_local, specular_alpha, UP);
        float odh_y = dot3(w_o_local, w_h_specular_local);
        float spec_i_dot_n_local = reflect3(-w_o_local, w_h_specular_local).z;
        float vis_y = spec_i_dot_n_local > 0.0f ? rp_fdiv(2.0f * spec_i_dot_n_local, rp_smith_den1(spec_i_dot_n_local, specular_alpha * specular_alpha)) : 0.0f;
        lobes = rp_gltf_component_sampler(m, odh_x, odh_y, 1.0f, vis_y);
        // glft_sample_reuse_component (:395-409), 2 components, only the index is used afterwards
        float rnd = fresnel_sample.x;
        float next_base = 0.0f;
        if (lobes.w0 > 0.0f && rnd >= next_base) component = 0;
        next_base += lobes.w0;
        if (lobes.w1 > 0.0f && rnd >= next_base) component = 1;
    }
    float cos_theta_h;
    if (component == 0) {
        w_i = w_i_diffuse;
        V3 w_h = norm3(w_i + w_o);
        cos_theta_h = dot3(n, w_h);
    } else {
        V3 w_h = w_h_specular_local;
        cos_theta_h = w_h.z;
        w_h = mul(frame, w_h);
        w_i = reflect3(-w_o, w_h);
    }
    float i_dot_n = dot3(n, w_i);
    if (!(i_dot_n > 0.0f)) {
        pdf = 0.0f;
        return v3s(0.0f);
    }
    pdf = RP_1_PI * fabsf(i_dot_n);
    if (m.ior > 1.0f) {
        pdf *= lobes.w0;
        float specular = rp_gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        pdf += specular * lobes.w1;
    }
    if (!(pdf > 0.0f)) return v3s(0.0f);
    V3 result = rp_gltf_bsdf(m, n, w_o, w_i);
    mis_wpdf = rp_gltf_wpdf(m, n, w_o, w_i);
    return result * fabsf(i_dot_n) / pdf;
}

// ------------------------------------------------------------------ glTF BSDF with the transmission lobe
// gltf_bsdf.glsl built with GLTF_SUPPORT_TRANSMISSION + GLTF_SUPPORT_TRANSMISSION_ROUGHNESS (RPTR_VARIANT_GLTF_TRANSMISSION): three
// components -- diffuse, GGX reflection, GGX transmission (refraction through ONESIDED surfaces, thin double reflection otherwise)
RP_DEV V3 rp_refract3(V3 I, V3 N, float eta) { // GLSL refract
    const float d = dot3(N, I);
    const float k = 1.0f - (eta * eta) * (1.0f - d * d);
    if (k < 0.0f) return v3s(0.0f);
    return eta * I - (eta * d + rp_fsqrt(k)) * N;
}
RP_DEV float rp_gltf_transmission_alpha(const RpMaterial &m) { return fmaxf(m.transmission_roughness * m.transmission_roughness, 0.002f); } // :278-282
RP_DEV V3 rp_gltf_t_bsdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) { // :294-359
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    float ior = o_dot_n < 0.0f ? rp_frcp(m.ior) : m.ior;
    const bool onesided = (m.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0u;
    V3 w_h;
    if (i_dot_n * o_dot_n < 0.0f) {
        if (!(m.specular_transmission > 0.f)) return v3s(0.0f);
        if (onesided)
            w_h = (-ior) * w_i - w_o;
        else
            w_h = reflect3(w_i, n) + w_o;
        if (!(dot3(w_h, n) > 0.0f)) return v3s(0.0f);
    } else
        w_h = w_i + w_o;
    w_h = norm3(w_h);
    float o_dot_h = dot3(w_o, w_h), i_dot_h = dot3(w_i, w_h);
    V3 diffuse = rp_gltf_diffuse_basecolor(m) * RP_1_PI;
    V3 specular = v3s(0.0f);
    if (m.ior > 1.0f) {
        V3 f0 = rp_gltf_specular_basecolor(m, m.ior);
        float specular_alpha = rp_gltf_specular_alpha(m);
        if (i_dot_n * o_dot_n < 0.0f) specular_alpha = rp_gltf_transmission_alpha(m);
        float specular_refl = rp_gtr_2(dot3(n, w_h), specular_alpha);
        specular_refl *= rp_smith_ggx(o_dot_n, i_dot_n, specular_alpha);
        float f_weight = rp_gltf_schlick_weight(fabsf(o_dot_h), ior);
        V3 F = mix3(f0, v3s(1.0f), f_weight);
        if (i_dot_n * o_dot_n < 0.0f) {
            diffuse = v3s(0.0f);
            specular = ((specular_refl * (1.f - m.metallic)) * m.specular_transmission) * m.transmission_color * (v3s(1.0f) - F);
            if (onesided) { // transmission angle compression
                float angle_compression = rp_fdiv(2.0f * o_dot_h, i_dot_h * ior + o_dot_h);
                specular = specular * (angle_compression * angle_compression);
            }
        } else {
            diffuse = diffuse * (1.0f - m.specular_transmission);
            diffuse = diffuse * (v3s(1.0f) - F);
            specular = specular_refl * F;
        }
    }
    return diffuse + specular;
}
struct RpLobes3 {
    float w0, w1, w2;
};
RP_DEV RpLobes3 rp_gltf_t_component_sampler(const RpMaterial &m, float ior, float odh_x, float odh_y, float odh_z, float vis_x, float vis_y, float vis_z) { // :366-394
    RpLobes3 c;
    float specular_base_lum = luminance3(rp_gltf_specular_basecolor(m, m.ior));
    float F0 = mixf(specular_base_lum, 1.0f, rp_gltf_schlick_weight(odh_x, 1.0f));
    float F1 = mixf(specular_base_lum, 1.0f, rp_gltf_schlick_weight(odh_y, 1.0f));
    float F2 = mixf(specular_base_lum, 1.0f, rp_gltf_schlick_weight(odh_z, ior));
    c.w0 = (1.0f - F0) * vis_x * (1.0f - m.metallic) * luminance3(rp_gltf_diffuse_basecolor(m));
    c.w1 = F1 * vis_y;
    c.w0 *= (1.0f - m.specular_transmission);
    c.w2 = (1.0f - F2) * vis_z * (1.0f - m.metallic) * m.specular_transmission;
    float weight_sum = 0.0f;
    weight_sum += c.w0;
    weight_sum += c.w1;
    weight_sum += c.w2;
    if (weight_sum > 0.0f) {
        c.w0 = rp_fdiv(c.w0, weight_sum);
        c.w1 = rp_fdiv(c.w1, weight_sum);
        c.w2 = rp_fdiv(c.w2, weight_sum);
    } else
        c.w0 = 1.0f;
    return c;
}
RP_DEV float rp_gltf_t_wpdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) { // :414-494
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    float ior = o_dot_n < 0.0f ? rp_frcp(m.ior) : m.ior;
    const bool onesided = (m.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0u;
    float pdf = RP_1_PI * fabsf(i_dot_n);
    if (m.ior > 1.0f) {
        V3 w_h;
        if (i_dot_n * o_dot_n < 0.0f) {
            if (!(m.specular_transmission > 0.f)) return 0.0f;
            if (onesided)
                w_h = (-ior) * w_i - w_o;
            else
                w_h = reflect3(w_i, n) + w_o;
            if (!(dot3(w_h, n) > 0.0f)) return 0.0f;
        } else
            w_h = w_i + w_o;
        w_h = norm3(w_h);
        float o_dot_h = dot3(w_o, w_h), i_dot_h = dot3(w_i, w_h);
        float cos_theta_h = dot3(w_h, n);
        float specular_alpha = rp_gltf_specular_alpha(m);
        float vis_y = rp_fdiv(2.0f * fabsf(i_dot_n), rp_smith_den1(i_dot_n, specular_alpha * specular_alpha));
        float vis_z = vis_y;
        float transmission_alpha = specular_alpha;
        if (m.specular_transmission > 0.f) {
            transmission_alpha = rp_gltf_transmission_alpha(m);
            vis_z = rp_fdiv(2.0f * fabsf(i_dot_n), rp_smith_den1(i_dot_n, transmission_alpha * transmission_alpha));
        }
        RpLobes3 c = rp_gltf_t_component_sampler(m, ior, fabsf(o_dot_h), fabsf(o_dot_h), fabsf(o_dot_h), 1.0f, vis_y, vis_z);
        if (i_dot_n * o_dot_n < 0.0f) specular_alpha = transmission_alpha;
        float specular = rp_gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        if (i_dot_n * o_dot_n < 0.0f) {
            if (onesided) {
                float angle_compression = rp_fdiv(2.0f * o_dot_h, i_dot_h * ior + o_dot_h);
                specular *= angle_compression * angle_compression;
            }
            pdf = specular * c.w2;
        } else {
            pdf *= c.w0;
            pdf += specular * c.w1;
        }
    }
    return pdf;
}
RP_DEV V3 rp_sample_gltf_t_brdf(const RpMaterial &m, V3 n, V3 w_o, V3 &w_i, float &pdf, float &mis_wpdf, V2 rng_sample, V2 fresnel_sample, V3 v_x,
                                V3 v_y) { // :496-645
    M3 frame{v_x, v_y, n};
    V3 w_o_local = mul_t(frame, w_o);
    float o_dot_n = w_o_local.z;
    float ior = o_dot_n < 0.0f ? rp_frcp(m.ior) : m.ior;
    const bool onesided = (m.flags & RPTR_BASE_MATERIAL_ONESIDED) != 0u;
    mis_wpdf = 0.0f;
    if (o_dot_n < 0.0f) w_o_local.z = -w_o_local.z;
    V3 UP = rp_to_pipe_sample(rng_sample);
    V3 w_i_diffuse = norm3(n + rp_sample_sphere(UP));
    if (o_dot_n < 0.0f) w_i_diffuse = -w_i_diffuse;
    float specular_alpha = rp_gltf_specular_alpha(m);
    int component = 0;
    RpLobes3 lobes{0.0f, 0.0f, 0.0f};
    V3 w_h_specular_local = v3s(0.0f), w_h_transmission_local = v3s(0.0f);
    if (m.ior > 1.0f) {
        float odh_x = rp_cos_half_angle(dot3(w_o, w_i_diffuse));
        w_h_specular_local = rp_sample_gtr_2_vndf(w_o_local, specular_alpha, UP);
        float odh_y = dot3(w_o_local, w_h_specular_local);
        float spec_i_dot_n_local = reflect3(-w_o_local, w_h_specular_local).z;
        float vis_y = spec_i_dot_n_local > 0.0f ? rp_fdiv(2.0f * spec_i_dot_n_local, rp_smith_den1(spec_i_dot_n_local, specular_alpha * specular_alpha)) : 0.0f;
        float transmission_alpha = specular_alpha;
        w_h_transmission_local = w_h_specular_local;
        float odh_z = odh_y;
        float trans_i_dot_n_local = spec_i_dot_n_local;
        float vis_z = 0.0f;
        if (m.specular_transmission > 0.f) {
            transmission_alpha = rp_gltf_transmission_alpha(m);
            w_h_transmission_local = rp_sample_gtr_2_vndf(w_o_local, transmission_alpha, UP);
            odh_z = dot3(w_o_local, w_h_transmission_local);
            if (onesided)
                trans_i_dot_n_local = -rp_refract3(-w_o_local, w_h_transmission_local, rp_frcp(ior)).z;
            else
                trans_i_dot_n_local = reflect3(-w_o_local, w_h_transmission_local).z;
            vis_z = trans_i_dot_n_local > 0.0f ? rp_fdiv(2.0f * trans_i_dot_n_local, rp_smith_den1(trans_i_dot_n_local, transmission_alpha * transmission_alpha)) : 0.0f;
        }
        lobes = rp_gltf_t_component_sampler(m, ior, odh_x, odh_y, odh_z, 1.0f, vis_y, vis_z);
        // glft_sample_reuse_component (:395-409), 3 components, only the index is used afterwards
        float rnd = fresnel_sample.x;
        float next_base = 0.0f;
        if (lobes.w0 > 0.0f && rnd >= next_base) component = 0;
        next_base += lobes.w0;
        if (lobes.w1 > 0.0f && rnd >= next_base) component = 1;
        next_base += lobes.w1;
        if (lobes.w2 > 0.0f && rnd >= next_base) component = 2;
    }
    float cos_theta_h, i_dot_h, o_dot_h;
    if (component == 0) {
        w_i = w_i_diffuse;
        V3 w_h = norm3(w_i + w_o);
        cos_theta_h = dot3(n, w_h);
        i_dot_h = o_dot_h = dot3(w_o, w_h);
    } else {
        if (component == 2) {
            specular_alpha = rp_gltf_transmission_alpha(m);
            w_h_specular_local = w_h_transmission_local;
        }
        V3 w_h = w_h_specular_local;
        if (o_dot_n < 0.0f) w_h.z = -w_h.z; // flip into the original frame if necessary
        cos_theta_h = w_h.z;
        w_h = mul(frame, w_h);
        i_dot_h = o_dot_h = dot3(w_o, w_h);
        if (component != 1) {
            if (onesided) {
                w_i = rp_refract3(-w_o, w_h, rp_frcp(ior));
                i_dot_h = dot3(w_i, w_h);
            } else
                w_i = reflect3(reflect3(-w_o, w_h), n);
        } else
            w_i = reflect3(-w_o, w_h);
    }
    float i_dot_n = dot3(n, w_i);
    if ((i_dot_n * o_dot_n > 0.0f) != (component != 2)) {
        pdf = 0.0f;
        return v3s(0.0f);
    }
    pdf = RP_1_PI * fabsf(i_dot_n);
    if (m.ior > 1.0f) {
        pdf *= lobes.w0;
        float specular = rp_gtr_2_vndf_pdf(o_dot_n, cos_theta_h, specular_alpha);
        if (i_dot_n * o_dot_n < 0.0f) {
            if (onesided) {
                float angle_compression = rp_fdiv(2.0f * o_dot_h, i_dot_h * ior + o_dot_h);
                specular *= angle_compression * angle_compression;
            }
            pdf = specular * lobes.w2;
        } else
            pdf += specular * lobes.w1;
    }
    if (!(pdf > 0.0f)) return v3s(0.0f);
    V3 result = rp_gltf_t_bsdf(m, n, w_o, w_i);
    mis_wpdf = rp_gltf_t_wpdf(m, n, w_o, w_i);
    return result * fabsf(i_dot_n) / pdf;
}

// ------------------------------------------------------------------ Lambert BSDF, rendering/bsdfs/simple_bsdf.glsl
RP_DEV V3 rp_simple_bsdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) { // :44-59
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    V3 diffuse = m.base_color * RP_1_PI;
    if (i_dot_n * o_dot_n < 0.0f) return v3s(0.0f);
    return diffuse;
}
RP_DEV float rp_simple_pdf(V3 n, V3 w_o, V3 w_i) { // :68-83
    float i_dot_n = dot3(n, w_i);
    float o_dot_n = dot3(n, w_o);
    float pdf = RP_1_PI * fabsf(i_dot_n);
    if (i_dot_n * o_dot_n < 0.0f) return 0.0f;
    return pdf;
}
RP_DEV V3 rp_sample_simple_brdf(const RpMaterial &m, V3 n, V3 &w_i, float &pdf, float &mis_pdf, V2 rnd) { // :61-66,85-94
    float phi = 2.0f * RP_PI * rnd.x;
    float cos_theta = rnd.y * 2.0f - 1.0f;
    float sin_theta = rp_fsqrt(1.0f - cos_theta * cos_theta);
    V3 sph = v3(sin_theta * cosf(phi), sin_theta * sinf(phi), cos_theta);
    w_i = norm3(n + sph);
    float i_dot_n = dot3(n, w_i);
    pdf = mis_pdf = RP_1_PI * fabsf(i_dot_n);
    return m.base_color;
}

// material registration (gltf_bsdf.glsl:649-655, simple_bsdf.glsl:98-104)
template <int VARIANT>
RP_DEV V3 rp_eval_bsdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) {
    if (VARIANT == RPTR_VARIANT_GLTF_TRANSMISSION) return rp_gltf_t_bsdf(m, n, w_o, w_i);
    return VARIANT == RPTR_VARIANT_SIMPLE ? rp_simple_bsdf(m, n, w_o, w_i) : rp_gltf_bsdf(m, n, w_o, w_i);
}
template <int VARIANT>
RP_DEV float rp_eval_bsdf_wpdf(const RpMaterial &m, V3 n, V3 w_o, V3 w_i) {
    if (VARIANT == RPTR_VARIANT_GLTF_TRANSMISSION) return rp_gltf_t_wpdf(m, n, w_o, w_i);
    return VARIANT == RPTR_VARIANT_SIMPLE ? rp_simple_pdf(n, w_o, w_i) : rp_gltf_wpdf(m, n, w_o, w_i);
}

// ------------------------------------------------------------------ triangle lights (a12), rendering/lights/tri.glsl
RP_DEV float rp_fast_positive_atan(float y) { // :58-74
    float rx, ry, rz;
    rx = (fabsf(y) > 1.0f) ? rp_frcp(fabsf(y)) : fabsf(y);
    ry = rx * rx;
    rz = fmaf(ry, 0.02083509974181652f, -0.08513300120830536f);
    rz = fmaf(ry, rz, 0.18014100193977356f);
    rz = fmaf(ry, rz, -0.3302994966506958f);
    ry = fmaf(ry, rz, 0.9998660087585449f);
    rz = fmaf(-2.0f * ry, rx, float(0.5f * RP_PI));
    rz = (fabsf(y) > 1.0f) ? rz : 0.0f;
    rx = fmaf(rx, ry, rz);
    return (y < 0.0f) ? (RP_PI - rx) : rx;
}
RP_DEV float rp_half_tri_solid_angle_tan(V3 v0, V3 v1, V3 v2, V3 &tp) { // :83-113
    float householder_sign = (v0.x > 0.0f) ? -1.0f : 1.0f;
    float hs = rp_frcp(fabsf(v0.x) + 1.0f);
    float hy = v0.y * hs, hz = v0.z * hs;
    float dot_0_1 = dot3(v0, v1);
    float dot_0_2 = dot3(v1, v2);
    float dot_1_2 = dot3(v0, v2);
    float dh0 = fmaf(-householder_sign, v1.x, dot_0_1);
    float dh2 = fmaf(-householder_sign, v2.x, dot_1_2);
    float m00 = fmaf(-dh0, hy, v1.y), m01 = fmaf(-dh0, hz, v1.z);
    float m10 = fmaf(-dh2, hy, v2.y), m11 = fmaf(-dh2, hz, v2.z);
    float simplex_volume = fabsf(m00 * m11 - m10 * m01);
    float dot_0_2_plus_1_2 = dot_0_2 + dot_1_2;
    float one_plus_dot_0_1 = 1.0f + dot_0_1;
    tp = v3(simplex_volume, dot_0_2_plus_1_2, one_plus_dot_0_1);
    return rp_fdiv(simplex_volume, one_plus_dot_0_1 + dot_0_2_plus_1_2);
}
RP_DEV V3 rp_sample_solid_angle_polygon(V3 v0, V3 v1, V3 v2, float solid_angle, V3 params, V2 rnd) { // :132-152
    float sub = solid_angle * rnd.x;
    V3 vert0 = v1, vert1 = v0, vert2 = v2;
    float cs = cosf(0.5f * sub), sn = sinf(0.5f * sub);
    V3 offset = vert0 * (params.x * cs - params.y * sn) + vert2 * (params.z * sn);
    float k = 2.0f * rp_fdiv(dot3(vert0, offset), dot3(offset, offset));
    V3 new_vertex_2 = v3(fmaf(k, offset.x, -vert0.x), fmaf(k, offset.y, -vert0.y), fmaf(k, offset.z, -vert0.z));
    float s2 = dot3(vert1, new_vertex_2);
    float s = rp_mix_fma(1.0f, s2, rnd.y);
    float denominator = fmaf(-s2, s2, 1.0f);
    float t_normed = rp_fsqrt(rp_fdiv(fmaf(-s, s, 1.0f), denominator));
    t_normed = (denominator > 0.0f) ? t_normed : rnd.y;
    return fmaf(-t_normed, s2, s) * vert1 + t_normed * new_vertex_2;
}
RP_DEV void rp_load_light(const RptrTriLightData *lights, int id, V3 &a, V3 &b, V3 &c, V3 &rad) {
    // 48 bytes = 3 x float4
    const float4 *p = reinterpret_cast<const float4 *>(lights + id);
    float4 q0 = p[0], q1 = p[1], q2 = p[2];
    a = v3(q0.x, q0.y, q0.z);
    b = v3(q0.w, q1.x, q1.y);
    c = v3(q1.z, q1.w, q2.x);
    rad = v3(q2.y, q2.z, q2.w);
}
// rendering/mc/lights_linear.glsl:19-127 (binned RIS, solid-angle sampling), in three parts so that the expensive one --
// the approximate contribution of every light of the chosen bin -- can be evaluated by any lane of the wave
// (kernels.h: the lanes of a wave share the candidates of all their tri-light samples; results are bit-identical to
// the straight loop because each contribution is the same arithmetic and the owner sums them in the same order).
struct RpLightBin { // :21-36
    int bin_begin, bin_end;
    float sel_p;
};
RP_DEV RpLightBin rp_choose_light_bin(const RpScene &sc, const RpFrame &f, float sel_x) {
    const int num_bins = f.num_bins;
    sel_x *= float(num_bins);
    int bin_id = int(uint32_t(sel_x));
    bin_id = min(bin_id, num_bins - 1);
    RpLightBin b;
    b.sel_p = rp_frcp(float(num_bins));
    b.bin_begin = f.lc.bin_size * bin_id;
    b.bin_end = min(f.lc.bin_size * (bin_id + 1), sc.num_lights);
    return b;
}
// :41-66 the approximate (unshadowed, solid-angle) contribution of one light to hit_p; 0 for ids beyond the bin
RP_DEV float rp_tri_light_contribution(const RpScene &sc, int light_id, int bin_end, V3 hit_p, V3 hit_n) {
    const float MIN_IRRADIANCE = 6.2e-4f * 0.001f;
    float contrib = 0.0f;
    if (light_id < bin_end) {
        V3 a, b, c, rad;
        rp_load_light(sc.lights, light_id, a, b, c, rad);
        a = a - hit_p;
        b = b - hit_p;
        c = c - hit_p;
        bool front_facing = dot3(cross3(a, b), c) < 0.0f; // tri.glsl:21-23
        contrib = luminance3(rad);
        if ((dot3(a, hit_n) > 0.0f || dot3(b, hit_n) > 0.0f || dot3(c, hit_n) > 0.0f) && front_facing) {
            a = norm3(a);
            b = norm3(b);
            c = norm3(c);
            V3 tp;
            contrib *= 2.0f * rp_fast_positive_atan(rp_half_tri_solid_angle_tan(a, b, c, tp));
        } else
            contrib = 0.0f;
        contrib += MIN_IRRADIANCE;
    }
    return contrib;
}
// :68-127 selection among the bin's contributions (read through `contribution(i)`, i = 0..15 in order) + solid-angle sample
template <class Contribution>
RP_DEV V3 rp_finish_tri_light_sample(const RpScene &sc, const RpFrame &f, const RpLightBin &bin, Contribution contribution, V3 hit_p, V2 dir_sample,
                                     float sel_y, V3 &light_dir, float &light_dist, float &pdf, float &mis_wpdf) {
    float total_contrib = 0.0f;
#pragma unroll
    for (int i = 0; i < RPTR_BINNED_LIGHTS_BIN_MAX_SIZE; ++i)
        if (bin.bin_begin + i < bin.bin_end) total_contrib += contribution(i);
    float p = 0.0f, t = 0.0f;
    int light_id = bin.bin_begin;
    bool done = false;
#pragma unroll
    for (int i = 0; i < RPTR_BINNED_LIGHTS_BIN_MAX_SIZE; ++i) {
        if (!done) {
            light_id = bin.bin_begin + i;
            if (!(light_id < bin.bin_end)) {
                done = true;
            } else {
                p = rp_fdiv(contribution(i), total_contrib);
                t += p;
                if (sel_y < t) done = true;
            }
        }
    }
    const float sel_p = bin.sel_p * p;
    V3 l0, l1, l2, lrad;
    rp_load_light(sc.lights, light_id, l0, l1, l2, lrad); // may be bin_end: zero-padded
    V3 d0 = norm3(l0 - hit_p);
    V3 d1 = norm3(l1 - hit_p);
    V3 d2 = norm3(l2 - hit_p);
    V3 tp;
    float polygon_solid_angle = 2.0f * rp_fast_positive_atan(rp_half_tri_solid_angle_tan(d0, d1, d2, tp));
    light_dir = rp_sample_solid_angle_polygon(d0, d1, d2, polygon_solid_angle, tp, dir_sample);
    pdf = rp_frcp(polygon_solid_angle);
    V3 e_n = cross3(l1 - l0, l2 - l0);
    light_dist = rp_fdiv(dot3(l0 - hit_p, e_n), dot3(light_dir, e_n));
    mis_wpdf = rp_fdiv(2.0f * light_dist * light_dist, fabsf(dot3(light_dir, e_n)));
    pdf *= sel_p;
    mis_wpdf = rp_fdiv(mis_wpdf, float(f.num_bins));
    return 1.0f * lrad / pdf;
}

// ------------------------------------------------------------------ sun + sky (a11, a15)
RP_DEV V3 rp_sample_sun_dir(V3 sun_dir, float cos_radius, V2 s) { // rendering/lights/sun.glsl:9-15
    float phi = 2.0f * RP_PI * s.x;
    float cosTheta = mixf(1.0f, cos_radius, s.y);
    float sinTheta = rp_fsqrt(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
    V3 vx, vy;
    rp_ortho_basis(vx, vy, sun_dir);
    M3 fr{vx, vy, sun_dir};
    return mul(fr, v3(sinTheta * cosf(phi), sinTheta * sinf(phi), cosTheta));
}
RP_DEV float rp_sun_dir_pdf(float cos_radius) { return rp_frcp(2.0f * RP_PI * (1.0f - cos_radius)); } // sun.glsl:17-20
RP_DEV float rp_nee_mis(float pdf_f, float pdf_g) { return rp_fdiv(pdf_f, pdf_f + pdf_g); }                 // nee_interface.glsl:11-15 (n=1)

// rendering/lights/sky_model_arhosek/sky_model.glsl:40-59
RP_DEV V3 rp_skymodel_radiance(const RptrSkyModelParams &st, V3 sun_dir, V3 view_dir) {
    float cosTheta = clamp1(view_dir.y, 0.0f, 1.0f);
    float cosGamma = clamp1(dot3(view_dir, sun_dir), -1.0f, 1.0f);
    float gamma = acosf(cosTheta);
#define RP_CFG(i) v3(st.configs[i][0], st.configs[i][1], st.configs[i][2])
    V3 c4g = RP_CFG(4) * gamma;
    V3 expM = v3(expf(c4g.x), expf(c4g.y), expf(c4g.z));
    float rayM = cosGamma * cosGamma;
    V3 c8 = RP_CFG(8);
    V3 mie_base = v3s(1.0f) + c8 * c8 - 2.0f * c8 * cosGamma;
    V3 mie_den = v3(mie_base.x * rp_fsqrt(mie_base.x), mie_base.y * rp_fsqrt(mie_base.y), mie_base.z * rp_fsqrt(mie_base.z));
    V3 mieM = v3s(1.0f + cosGamma * cosGamma) / mie_den;
    float zenith = rp_fsqrt(cosTheta);
    V3 c1d = RP_CFG(1) / (cosTheta + 0.01f);
    V3 e1 = v3(expf(c1d.x), expf(c1d.y), expf(c1d.z));
    V3 coeffs = (v3s(1.0f) + RP_CFG(0) * e1) * ((((RP_CFG(2) + RP_CFG(3) * expM) + RP_CFG(5) * rayM) + RP_CFG(6) * mieM) + RP_CFG(7) * zenith);
#undef RP_CFG
    return coeffs * v3(st.radiances[0], st.radiances[1], st.radiances[2]) * 0.01f;
}
// vulkan/pt_megakernel.glsl:113-149
RP_DEV V3 rp_compute_sky_illum(const RpFrame &f, V3 ray_dir, float prev_bsdf_pdf) {
    V3 sun_dir = ld3(f.sp.sun_dir);
    V3 dir = ray_dir;
    float ocean_coeff = 1.0f;
    if (dir.y <= 0.0f) {
        dir.y = -dir.y;
        ocean_coeff = 0.7f * pow5f(fmaxf(1.0f - fabsf(dir.y), 0.0f));
    }
    V3 atmosphere = max3(rp_skymodel_radiance(f.sp.sky_params, sun_dir, dir), v3s(0.0f)) * ocean_coeff;
    V3 sun_illum = v3s(0.0f);
    if (dot3(dir, sun_dir) >= f.sp.sun_cos_angle) sun_illum = ld3(f.sp.sun_radiance) * ocean_coeff;
    V3 illum = v3s(0.0f);
    illum = illum + abs3(atmosphere);
    float light_pdf = f.sp.sun_radiance[3] * rp_sun_dir_pdf(f.sp.sun_cos_angle);
    float w = rp_nee_mis(prev_bsdf_pdf, light_pdf);
    illum = illum + w * abs3(sun_illum);
    return illum;
}
// vulkan/geometry.glsl:76-78
RP_DEV float rp_geometry_scale_to_tmin(V3 orig, float geometry_scale) { return (len3_ieee(orig) + geometry_scale) * RPTR_RAY_EPSILON; } // (IEEE in both builds: dmath.h)
