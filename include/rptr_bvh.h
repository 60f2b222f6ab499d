/*
 * rptr_bvh.h -- in-memory layout of the software acceleration structure that
 * replaces the reference's driver-built BLAS/TLAS (vulkan/vulkanrt_utils.h:55-187).
 * The layout is part of the diagnostic ABI (rptr_hip_export_bvh) so that the
 * test oracle can walk the very same tree; see DESIGN.md "Data layout in HBM".
 *
 * Two-level BVH2:
 *   - one node array holds the TLAS (root = node 0) followed by every BLAS;
 *   - a node stores the AABBs of BOTH children (Aila/Laine style), so one
 *     64-byte fetch decides both slab tests;
 *   - child >= 0 : inner node index (absolute, into the shared node array)
 *     child <  0 : leaf, packed so that it can sit on the traversal stack as is:
 *                  v = -2 - child; first = v >> 3; count = v & 7
 *                  (RPTR_BVH_LEAF(first,count) / RPTR_BVH_LEAF_FIRST / _COUNT);
 *                  cnt0/cnt1 repeat the count. In the TLAS a leaf lists `count`
 *                  (0 or 1) RptrBvhInstance records, in a BLAS `count` RptrBvhTri.
 *   - an empty child (count == 0 leaf) has an inverted box (lo=+inf, hi=-inf).
 */
#ifndef RPTR_BVH_H
#define RPTR_BVH_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RPTR_BVH_MAX_LEAF_TRIS 4 /* must stay <= 7 (3 count bits) */
#define RPTR_BVH_LEAF(first, count) (-2 - (int32_t)((uint32_t)(first) * 8u + (uint32_t)(count)))
#define RPTR_BVH_LEAF_FIRST(child) ((int32_t)((uint32_t)(-2 - (child)) >> 3))
#define RPTR_BVH_LEAF_COUNT(child) ((int32_t)((uint32_t)(-2 - (child)) & 7u))
#define RPTR_BVH_STACK_DEPTH 64

typedef struct RptrBvhNode { /* 64 bytes */
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    int32_t cnt0, cnt1;
} RptrBvhNode;

/* Moeller-Trumbore ready triangle, object space of its mesh: 48 bytes */
typedef struct RptrBvhTri {
    float v0[3];
    float e1[3]; /* v1 - v0 */
    float e2[3]; /* v2 - v0 */
    uint32_t prim;     /* primitive index inside its geometry                    */
    uint32_t geom;     /* geometry index inside its mesh (rayQuery GeometryIndex) */
    uint32_t _pad;
} RptrBvhTri;

/* 128 bytes */
typedef struct RptrBvhInstance {
    float world_to_object[12]; /* row-major 3x4                                  */
    float object_to_world[12]; /* row-major 3x4                                  */
    int32_t blas_root;         /* absolute node index of the mesh's root         */
    int32_t geometry_base;     /* instanceCustomIndex = render_mesh_base_offset  */
    int32_t instance_id;       /* rayQueryGetIntersectionInstanceIdEXT            */
    int32_t flags;
    int32_t _pad[4];
} RptrBvhInstance;

#ifdef __cplusplus
}
#endif
#endif
