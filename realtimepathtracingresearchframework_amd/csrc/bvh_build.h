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
// Collapses the binary tree: a node adopts its grandchildren, largest box first, until it has four
// children (Wald et al. style). Leaves and the primitive order are kept.
void collapse_bvh4(const BuiltTree &tree, Wide4Tree &out);

} // namespace rptr
