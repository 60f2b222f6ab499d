// bvh_build.h -- host-side construction of the acceleration structure: binary binned-SAH build + 4-wide collapse.
//
// Stands in for the driver's vkCmdBuildAccelerationStructuresKHR that the
// reference calls through vulkan/vulkanrt_utils.h:83-105 (BLAS per mesh, static
// meshes with PREFER_FAST_TRACE: render_vulkan.cpp:942-952; TLAS over instances:
// :1219-1321). Binned SAH, multi-threaded over subtrees; output is the flat
// two-children-per-node layout of include/rptr_bvh.h.
#pragma once
#include "../../include/rptr_bvh.h"
#include "bvh4.h"
#include <cstdint>
#include <vector>

namespace rptr {

struct BuildPrim {
    float lo[3], hi[3];
};

struct BuiltTree {
    std::vector<RptrBvhNode> nodes; // root = nodes[0]; child indices relative to this vector
    std::vector<uint32_t> order;    // leaf order -> input primitive index; leaves reference ranges of it
    float lo[3], hi[3];             // root bounds
    int depth = 0;
};

// max_leaf: maximum primitives per leaf; max_depth: hard bound on tree depth
// (median splits below it). threads <= 0 -> hardware_concurrency.
void build_bvh2(const BuildPrim *prims, uint32_t n, uint32_t max_leaf, int max_depth, int threads, BuiltTree &out);

// The same output from PLOC (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume Hierarchy Construction", TVCG
// 2018): clusters in Morton order, every cluster finds the neighbour within `radius` positions whose union with it has the smallest
// area, mutual nearest neighbours merge, repeat; leaves by SAH over the finished tree. The host statement of what the device builder
// (csrc/ploc.h) computes -- same order of operations, so both give the same tree.
void build_bvh2_ploc(const BuildPrim *prims, uint32_t n, int radius, uint32_t max_leaf, int threads, BuiltTree &out);

// bottom-up refit of node boxes from new primitive bounds given in LEAF order
// (prims[i] corresponds to order[i]); host fallback used by tests.
void refit_bvh2(BuiltTree &tree, const BuildPrim *prims_leaf_order);

// ---- 4-wide form (what the device traverses)
struct Wide4 {
    RpBox4 box;       // float boxes of the children
    int32_t child[4]; // >= 0: index into Wide4Tree::nodes; <= -2: packed leaf (first relative to `order`); RPTR_BVH4_EMPTY
};
struct Wide4Tree {
    std::vector<Wide4> nodes; // root = nodes[0], breadth-first order
    float lo[3], hi[3];
};
// Collapses the binary tree into nodes of up to four children. Leaves and the primitive order are kept. Rules:
//   COLLAPSE_OPTIMAL (default of the host builder): the collapse that minimises the summed surface area of the binary nodes that survive as
//     wide nodes -- each survivor is a node visit with probability ~ its area, the leaves are the same whatever the collapse -- by dynamic
//     programming over "subtree c shown through at most i slots of its parent" (Ylitie, Karras, Laine, "Efficient Incoherent Ray Traversal on
//     GPUs Through Compressed Wide BVHs", HPG 2017, section 3.1, here for four slots). Against the greedy rule: a third fewer nodes on the
//     height field (299 k -> 192 k), 1-4 % fewer node visits per ray on every scene;
//   COLLAPSE_GREEDY: a node adopts its grandchildren, largest box first, until it has four children (Wald et al. style): rounds 1-3; kept
//     for the top level over instance boxes and for the meshes of re-braided scenes (the device builder, csrc/ploc.h, implements
//     COLLAPSE_OPTIMAL, and so does its host statement build_bvh2_ploc + collapse_bvh4);
//   COLLAPSE_EVEN: every inner child hands its two children up (the device's rebuild, csrc/lbvh.h).
// rule < 0: RPTR_COLLAPSE = optimal | dp | greedy | even, default `fallback`.
enum { COLLAPSE_GREEDY = 0, COLLAPSE_EVEN = 1, COLLAPSE_OPTIMAL = 2 };
// Builder experiments (library options "collapse", "ploc_top", "ploc_leaf", include/rptr_hip.h; the caller -- build_host_bvh in rptr_hip.hip --
// sets them from the handle's options before it builds; nothing here reads the environment). collapse_rule < 0: the caller's fallback rule;
// ploc_top 0: RP_PLOC_TOP_DEFAULT; ploc_leaf 0: leaves by SAH.
struct BuildTuning {
    int collapse_rule = -1;
    size_t ploc_top = 0;
    int ploc_leaf = 0;
};
BuildTuning &build_tuning();
// clusters at which PLOC stops and a binned-SAH tree takes over: ONE default for the device builder (csrc/ploc.h RP_PLOC_TOP) and its host
// statement (build_bvh2_ploc), so that "the same tree" does not depend on a test setting RPTR_PLOC_TOP
#define RP_PLOC_TOP_DEFAULT 65536
void collapse_bvh4(const BuiltTree &tree, Wide4Tree &out, int rule = -1, int fallback = COLLAPSE_OPTIMAL);

} // namespace rptr

namespace rptr {

// ---- triangle pre-splitting (spatial splits before the build; static geometry only)
// A long thin diagonal triangle has a box that is almost entirely empty; a tree over such boxes overlaps everywhere (the forest of
// config C4: 7.5 triangle tests and 26 node visits per closest-hit ray). The triangle is therefore REFERENCED several times, each
// reference with the box of the part of the triangle inside one cell of a hierarchical grid (Karras & Aila 2013, section 4: split
// planes are the most important median planes of the scene's cubic grid that cross the box, the number of splits of a triangle is
// floor(density * cbrt(2^-level * (box area - ideal area) / scene extent^2)) -- an absolute rule, so that a mesh of well-shaped
// triangles is left alone -- scaled down when more than `budget` * n extra references would come out).
// References of one triangle all name the same triangle record: a hit has the same t / u / v / ids whichever reference finds it, so the
// closest-hit answer (smallest t, ties by (instance, geometry, primitive)) does not depend on the splitting. Boxes are computed in
// double precision and rounded outwards, so the references of a triangle cover it completely.
struct TriVerts {
    float v[3][3];
};
// out_box[r] / out_tri[r]: reference r and the triangle it belongs to (every triangle gets at least one reference; order: by triangle)
void presplit_triangles(const TriVerts *tris, uint32_t n, float density, float budget, int max_refs_per_tri, int threads, std::vector<BuildPrim> &out_box,
                        std::vector<uint32_t> &out_tri);

} // namespace rptr
