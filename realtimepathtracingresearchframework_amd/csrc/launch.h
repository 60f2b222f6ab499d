// launch.h -- host-side launchers of the path stages. The stages are templates (kernels.h) with a few hundred instantiations between
// them; each k_*.hip instantiates one family and exports a plain function that maps run-time flags to the instantiation, so that the
// families compile side by side (build.py) and rptr_hip.hip never instantiates a path kernel itself.
#pragma once
#include <hip/hip_ext.h>
#include <type_traits>

#include "kernels.h"

struct RpLaunch {
    unsigned grid;
    hipStream_t stream;
    hipEvent_t start, stop; // non-NULL: start / stop events ride on the dispatch packet itself (hipExtLaunchKernelGGL)
};
template <class K, class... A>
static inline void rp_launch_kernel(const RpLaunch &l, K kernel, unsigned block, A... args) {
    if (l.start)
        hipExtLaunchKernelGGL(kernel, dim3(l.grid), dim3(block), 0, l.stream, l.start, l.stop, 0, args...);
    else
        hipLaunchKernelGGL(kernel, dim3(l.grid), dim3(block), 0, l.stream, args...);
}
// run-time flag -> template argument: f(std::true_type) or f(std::false_type)
template <class F>
static inline void rp_pick(bool v, F &&f) {
    if (v)
        f(std::true_type());
    else
        f(std::false_type());
}

// k_extend.hip: closest-hit and shadow queries
void rp_launch_extend(const RpLaunch &l, bool count, bool first, bool alpha, bool single, bool table, const RpScene &sc, const RpFrame &f, const RpPathState &ps,
                      const uint32_t *queue, RpBounceCounters *bc, RpCounters *ctr, int *gstack);
void rp_launch_connect(const RpLaunch &l, bool count, bool alpha, bool single, const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq,
                       RpBounceCounters *bc, RpCounters *ctr, int *gstack);
hipError_t rp_extend_blocks_per_cu(int *out);
hipError_t rp_connect_blocks_per_cu(int single, int *out);
hipError_t rp_extend_later_blocks_per_cu(int *out);
#ifdef RP_PROF
hipError_t rp_prof_exchange(unsigned long long out[16]); // reads and clears the -DRP_PROF counters of rp_k_extend / rp_k_connect
#endif // occupancy of the traversal kernels (they all fit the same budget)

// k_shade.hip / k_tail.hip, built once per gpu-program variant (-DRP_INST_VARIANT=RPTR_VARIANT_*) and build of the shading arithmetic
// (-DRP_FAST_MATH=0|1, dmath.h: IEEE division / square root, or the hardware's 1-ulp reciprocal / square root; option "fast_math")
#define RP_DECLARE_VARIANT(V)                                                                                                                          \
    void rp_launch_shade_##V(const RpLaunch &l, bool first, bool lights, bool tex, bool table, const RpScene &sc, const RpFrame &f, const RpPathState &ps,  \
                              const RpShadowRays &sq, const uint32_t *order, const uint32_t *count_ptr, uint32_t *next_queue, uint32_t *next_count,     \
                              uint32_t *shadow_count, RpCounters *ctr);                                                                                  \
    void rp_launch_tail_##V(const RpLaunch &l, bool lights, bool full, bool single, bool table, const RpScene &sc, const RpFrame &f, const RpPathState &ps,  \
                             const RpShadowRays &sq, const uint32_t *queue, RpCounters *ctr, int first_bounce, int *gstack);
RP_DECLARE_VARIANT(ieee_v0)
RP_DECLARE_VARIANT(ieee_v1)
RP_DECLARE_VARIANT(ieee_v2)
RP_DECLARE_VARIANT(fast_v0)
RP_DECLARE_VARIANT(fast_v1)
RP_DECLARE_VARIANT(fast_v2)
#undef RP_DECLARE_VARIANT

template <class... A>
static inline void rp_launch_shade(int variant, bool fast_math, const RpLaunch &l, A... args) {
    if (variant == RPTR_VARIANT_SIMPLE)
        fast_math ? rp_launch_shade_fast_v1(l, args...) : rp_launch_shade_ieee_v1(l, args...);
    else if (variant == RPTR_VARIANT_GLTF_TRANSMISSION)
        fast_math ? rp_launch_shade_fast_v2(l, args...) : rp_launch_shade_ieee_v2(l, args...);
    else
        fast_math ? rp_launch_shade_fast_v0(l, args...) : rp_launch_shade_ieee_v0(l, args...);
}
template <class... A>
static inline void rp_launch_tail(int variant, bool fast_math, const RpLaunch &l, A... args) {
    if (variant == RPTR_VARIANT_SIMPLE)
        fast_math ? rp_launch_tail_fast_v1(l, args...) : rp_launch_tail_ieee_v1(l, args...);
    else if (variant == RPTR_VARIANT_GLTF_TRANSMISSION)
        fast_math ? rp_launch_tail_fast_v2(l, args...) : rp_launch_tail_ieee_v2(l, args...);
    else
        fast_math ? rp_launch_tail_fast_v0(l, args...) : rp_launch_tail_ieee_v0(l, args...);
}
static_assert(RPTR_VARIANT_GLTF == 0 && RPTR_VARIANT_SIMPLE == 1 && RPTR_VARIANT_GLTF_TRANSMISSION == 2, "k_shade.hip / k_tail.hip are built per variant number");
