/*
 * rptr_bvh.h -- in-memory layout of the software acceleration structure that
 * replaces the reference's driver-built BLAS/TLAS (vulkan/vulkanrt_utils.h:55-187).
 * The layout is part of the diagnostic ABI (rptr_hip_export_bvh) so that the
 * test oracle can walk the very same tree; see DESIGN.md "Data layout in HBM".
 *
 * Two-level, 4-wide, compressed (what the device traverses: RptrBvh4Node):
 *   - one node array holds the TLAS (root = node 0) followed by every BLAS;
 *   - a 64-byte node holds the boxes of up to FOUR children, quantised to 8 bits per plane on a
 *     per-node grid: plane = origin[a] + q * 2^(exp[a]-127). One 64-byte fetch decides four slab
 *     tests (the traversal is bound by the number of divergent fetches);
 *   - child >= 0 : inner node index (absolute, into the shared node array)
 *     child <= -2: leaf, packed so that it can sit on the traversal stack as is:
 *                  v = -2 - child; first = v >> 3; count = v & 7
 *                  (RPTR_BVH_LEAF(first,count) / RPTR_BVH_LEAF_FIRST / _COUNT).
 *                  In the TLAS a leaf lists `count` (1) RptrBvhInstance records, in a BLAS
 *                  `count` (1..RPTR_BVH_MAX_LEAF_TRIS) RptrBvhTri;
 *     child == RPTR_BVH4_EMPTY: unused slot (its box is inverted: qlo = 255, qhi = 0).
 *   - slots (0,1) and (2,3) are PAIRS for the traversal's visit order (csrc/dtraverse.h): siblings of the binary tree the node was
 *     collapsed from wherever the collapse was balanced.
 *
 * RptrBvhNode (2-wide, float boxes of both children) is the intermediate form of the host builder
 * (csrc/bvh_build.h) and of the test oracle's own tree; the device never sees it.
 */
#ifndef RPTR_BVH_H
#define RPTR_BVH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef RPTR_BVH_MAX_LEAF_TRIS
#define RPTR_BVH_MAX_LEAF_TRIS 4
#endif /* must stay <= 7 (3 count bits) */
#define RPTR_BVH_LEAF(first, count) (-2 - (int32_t)((uint32_t)(first) * 8u + (uint32_t)(count)))
#define RPTR_BVH_LEAF_FIRST(child) ((int32_t)((uint32_t)(-2 - (child)) >> 3))
#define RPTR_BVH_LEAF_COUNT(child) ((int32_t)((uint32_t)(-2 - (child)) & 7u))
#define RPTR_BVH_STACK_DEPTH 128 /* spill entries per thread behind the LDS part of the traversal stack */

typedef struct RptrBvhNode { /* 64 bytes */
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t cnt0, cnt1;
} RptrBvhNode;

#define RPTR_BVH4_EMPTY (INT32_MIN + 2)
typedef struct RptrBvh4Node { /* 64 bytes, four 16-byte loads */
    float origin[3];   /* lower corner of the node's box                                         */
    uint8_t exp[3];    /* biased float exponents of the grid steps: step[a] = 2^(exp[a]-127)     */
    uint8_t _pad0;
    uint8_t qlo[3][4]; /* [axis][child]: lower planes, rounded down                              */
    uint8_t qhi[3][4]; /* [axis][child]: upper planes, rounded up                                */
    int32_t child[4];
    uint32_t _pad1[2];
} RptrBvh4Node;

/* Moeller-Trumbore ready triangle, object space of its mesh: 48 bytes */
typedef struct RptrBvhTri {
    float v0[3];
    float e1[3]; /* v1 - v0 */
    float e2[3]; /* v2 - v0 */
    uint32_t prim;     /* primitive index inside its geometry                    */
    uint32_t geom;     /* geometry index inside its mesh (rayQuery GeometryIndex) */
    uint32_t flags;    /* bit 0, RPTR_BVH_TRI_ALPHA: some parameterized mesh gives this triangle a material without
                          BASE_MATERIAL_NOALPHA, i.e. a hit is a candidate for the alpha test (pt_megakernel.glsl:153-212);
                          bits 8..31: 0, or (flattened scenes, RPTR_FLATTEN) the index of the triangle's own instance record
                          in the instance array: the triangle is stored in world space and a hit belongs to that instance,
                          not to the one being traversed */
} RptrBvhTri;
#define RPTR_BVH_TRI_ALPHA 1u
#define RPTR_BVH_TRI_INSTANCE(flags) ((uint32_t)(flags) >> 8)
#define RPTR_BVH_INSTANCE_FLAT 1 /* RptrBvhInstance.flags: the one record a flattened scene's top level refers to */
#define RPTR_BVH_INSTANCE_OWN_MATERIALS 2 /* ... an instance of a parameterized mesh other than the FIRST one of its mesh: the per-triangle shading
                                            records of the mesh carry the first one's material ids, this instance resolves its own through its
                                            geometry records (csrc/dshade.h RpShadeTri) */

/* 128 bytes */
typedef struct RptrBvhInstance {
    float world_to_object[12]; /* row-major 3x4                                  */
    int32_t blas_root;         /* absolute node index of the mesh's root         */
    int32_t geometry_base;     /* instanceCustomIndex = render_mesh_base_offset  */
    int32_t instance_id;       /* rayQueryGetIntersectionInstanceIdEXT            */
    int32_t flags;             /* (the traversal reads the first 64 bytes)       */
    float object_to_world[12]; /* row-major 3x4 (refit: instance bounds)         */
    int32_t _pad[4];
} RptrBvhInstance;

#ifdef __cplusplus
}
#endif
#endif
