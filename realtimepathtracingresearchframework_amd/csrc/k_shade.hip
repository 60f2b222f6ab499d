// k_shade.hip -- instantiations of the shade stage (kernels.h rp_k_shade) for ONE gpu-program variant and ONE build of the shading arithmetic: -DRP_INST_VARIANT=0|1|2 -DRP_FAST_MATH=0|1 (dmath.h)
#include "launch.h"

#ifndef RP_INST_VARIANT
#error "build with -DRP_INST_VARIANT=<RPTR_VARIANT_*>"
#endif
#define RP_CAT2(a, b) a##b
#define RP_CAT(a, b) RP_CAT2(a, b)
#if RP_FAST_MATH
#define RP_LAUNCHER(stem) RP_CAT(RP_CAT(stem, fast_v), RP_INST_VARIANT)
#else
#define RP_LAUNCHER(stem) RP_CAT(RP_CAT(stem, ieee_v), RP_INST_VARIANT)
#endif

void RP_LAUNCHER(rp_launch_shade_)(const RpLaunch &l, bool first, bool lights, bool tex, bool table, const RpScene &sc, const RpFrame &f,
                                                const RpPathState &ps, const RpShadowRays &sq, const uint32_t *order, const uint32_t *count_ptr,
                                                uint32_t *next_queue, uint32_t *next_count, uint32_t *shadow_count, RpCounters *ctr) {
    rp_pick(first, [&](auto F) {
        rp_pick(lights, [&](auto L) {
            rp_pick(tex, [&](auto X) {
                rp_pick(table, [&](auto T) {
                    rp_launch_kernel(l, rp_k_shade<RP_INST_VARIANT, decltype(F)::value, decltype(L)::value, decltype(X)::value, decltype(T)::value>, 256u, sc, f, ps,
                                     sq, order, count_ptr, next_queue, next_count, shadow_count, ctr);
                });
            });
        });
    });
}
