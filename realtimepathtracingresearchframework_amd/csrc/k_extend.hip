// k_extend.hip -- instantiations of the traversal stages (kernels.h rp_k_extend, rp_k_connect) and their launchers (launch.h)
#include "launch.h"

void rp_launch_extend(const RpLaunch &l, bool count, bool first, bool alpha, bool single, bool table, const RpScene &sc, const RpFrame &f, const RpPathState &ps,
                      const uint32_t *queue, RpBounceCounters *bc, RpCounters *ctr, int *gstack) {
    // the point set only enters through the camera rays (FIRST) and the alpha test's generator (ALPHA)
    const bool t = table && (first || alpha);
    if (sc.lds_top && !count && !alpha && single && !t) { // RPTR_LDS_TOP=1: the instantiation with the top of the tree staged in LDS
        if (first)
            rp_launch_kernel(l, rp_k_extend_ldstop<true>, RP_TRAVERSE_BLOCK, sc, f, ps, queue, bc, ctr, gstack);
        else
            rp_launch_kernel(l, rp_k_extend_ldstop<false>, RP_TRAVERSE_BLOCK, sc, f, ps, queue, bc, ctr, gstack);
        return;
    }
    rp_pick(count, [&](auto C) {
        rp_pick(first, [&](auto F) {
            rp_pick(alpha, [&](auto A) {
                rp_pick(single, [&](auto S) {
                    rp_pick(t, [&](auto T) {
                        rp_launch_kernel(l, rp_k_extend<decltype(C)::value, decltype(F)::value, decltype(A)::value, decltype(S)::value, decltype(T)::value>,
                                         RP_TRAVERSE_BLOCK, sc, f, ps, queue, bc, ctr, gstack);
                    });
                });
            });
        });
    });
}

void rp_launch_connect(const RpLaunch &l, bool count, bool alpha, bool single, const RpScene &sc, const RpFrame &f, const RpPathState &ps, const RpShadowRays &sq,
                       RpBounceCounters *bc, RpCounters *ctr, int *gstack) {
    if (sc.lds_top && !count && !alpha && single) {
        rp_launch_kernel(l, rp_k_connect_ldstop<RP_LDS_TOP_NODES>, RP_TRAVERSE_BLOCK, sc, f, ps, sq, bc, ctr, gstack);
        return;
    }
    rp_pick(count, [&](auto C) {
        rp_pick(alpha, [&](auto A) {
            rp_pick(single, [&](auto S) {
                rp_launch_kernel(l, rp_k_connect<decltype(C)::value, decltype(A)::value, decltype(S)::value>, RP_TRAVERSE_BLOCK, sc, f, ps, sq, bc, ctr, gstack);
            });
        });
    });
}

hipError_t rp_extend_blocks_per_cu(int *out) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, rp_k_extend<false, true, false, false, false>, RP_TRAVERSE_BLOCK, 0);
}
hipError_t rp_connect_blocks_per_cu(int single, int *out) {
    return single ? hipOccupancyMaxActiveBlocksPerMultiprocessor(out, rp_k_connect<false, false, true>, RP_TRAVERSE_BLOCK, 0)
                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(out, rp_k_connect<false, false, false>, RP_TRAVERSE_BLOCK, 0);
}
hipError_t rp_extend_later_blocks_per_cu(int *out) {
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(out, rp_k_extend<false, false, false, false, false>, RP_TRAVERSE_BLOCK, 0);
}

#ifdef RP_PROF
hipError_t rp_prof_exchange(unsigned long long out[16]) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(rp_prof), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    unsigned long long zero[16] = {0};
    return hipMemcpyToSymbol(HIP_SYMBOL(rp_prof), zero, sizeof(zero));
}
#endif
