// k_tail.hip -- instantiations of the late-bounce kernel (kernels.h rp_k_tail) for ONE gpu-program variant and ONE build of the shading arithmetic: -DRP_INST_VARIANT=0|1|2 -DRP_FAST_MATH=0|1 (dmath.h)
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

// full: one instantiation serves textured and alpha-tested scenes (TEX = ALPHA = full)
void RP_LAUNCHER(rp_launch_tail_)(const RpLaunch &l, bool lights, bool full, bool single, bool table, const RpScene &sc, const RpFrame &f,
                                               const RpPathState &ps, const RpShadowRays &sq, const uint32_t *queue, RpCounters *ctr, int first_bounce,
                                               int *gstack) {
    rp_pick(lights, [&](auto L) {
        rp_pick(full, [&](auto F) {
            rp_pick(single, [&](auto S) {
                rp_pick(table, [&](auto T) {
                    rp_launch_kernel(l, rp_k_tail<RP_INST_VARIANT, decltype(L)::value, decltype(F)::value, decltype(F)::value, decltype(S)::value, decltype(T)::value>,
                                     256u, sc, f, ps, sq, queue, ctr, first_bounce, gstack);
                });
            });
        });
    });
}
