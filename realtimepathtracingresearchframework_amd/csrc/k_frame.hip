// k_frame.hip -- instantiations of the one-launch frame kernel (kernels.h rp_k_frame) for ONE gpu-program variant: -DRP_INST_VARIANT=0|1|2
#include "launch.h"

#ifndef RP_INST_VARIANT
#error "build with -DRP_INST_VARIANT=<RPTR_VARIANT_*>"
#endif
#define RP_CAT2(a, b) a##b
#define RP_CAT(a, b) RP_CAT2(a, b)

// full: one instantiation serves textured and alpha-tested scenes (TEX = ALPHA = full), as for the tail kernel
void RP_CAT(rp_launch_frame_v, RP_INST_VARIANT)(const RpLaunch &l, bool lights, bool full, bool single, bool table, const RpScene &sc, const RpFrame &f,
                                                const RpPathState &ps, const RpShadowRays &sq, const RpFrameQueues &fq, RpCounters *ctr, int *gstack) {
    rp_pick(lights, [&](auto L) {
        rp_pick(full, [&](auto F) {
            rp_pick(single, [&](auto S) {
                rp_pick(table, [&](auto T) {
                    rp_launch_kernel(l, rp_k_frame<RP_INST_VARIANT, decltype(L)::value, decltype(F)::value, decltype(F)::value, decltype(S)::value, decltype(T)::value>,
                                     256u, sc, f, ps, sq, fq, ctr, gstack);
                });
            });
        });
    });
}

// blocks of the frame kernel a CU holds (its LDS arena and VGPR budget; the grid is that many per CU: persistent blocks)
hipError_t RP_CAT(rp_frame_blocks_per_cu_v, RP_INST_VARIANT)(bool lights, bool full, int *out) {
    hipError_t e = hipSuccess;
    rp_pick(lights, [&](auto L) {
        rp_pick(full, [&](auto F) {
            e = hipOccupancyMaxActiveBlocksPerMultiprocessor(
                out, rp_k_frame<RP_INST_VARIANT, decltype(L)::value, decltype(F)::value, decltype(F)::value, true, false>, 256, 0);
        });
    });
    return e;
}

// the shader of the streaming frame (kernels.h rp_k_stream_shade)
void RP_CAT(rp_launch_stream_shade_v, RP_INST_VARIANT)(const RpLaunch &l, bool lights, bool tex, bool table, const RpScene &sc, const RpFrame &f, const RpPathState &ps,
                                                       const RpShadowRays &sq, const RpStream &sx, RpCounters *ctr) {
    rp_pick(lights, [&](auto L) {
        rp_pick(tex, [&](auto X) {
            rp_pick(table, [&](auto T) {
                rp_launch_kernel(l, rp_k_stream_shade<RP_INST_VARIANT, decltype(L)::value, decltype(X)::value, decltype(T)::value>, 256u, sc, f, ps, sq, sx, ctr);
            });
        });
    });
}
