// Microbenchmark: VALU issue rate of one SIMD of gfx950 (MI355X) -- wave64 instructions per clock per SIMD for independent streams
// of the instructions the BVH node step is made of, at 1..8 waves per SIMD. Settles the ceiling bench.py's `roofline.valu` uses
// (MI355X_MICROARCH.md says a wave64 VALU instruction issues over 2 cycles on SIMD-32; round 2 assumed 4).
//
// Every block is 256 threads = 4 waves = one wave per SIMD of its CU; W blocks per CU are forced by a dynamic LDS allocation of
// 160 KiB / W, so "W waves per SIMD" is exact when the dispatcher spreads the blocks (checked through the per-block cycle counts:
// a CU that got more than its share shows a longer s_memtime span). A wave runs ITER iterations of an unrolled block of UNROLL
// independent instructions over 8 accumulators (dependency distance 8 instructions >= the 4-cycle dependent latency at 2 cycles each).
// Reported: (a) chip-wide G wave-instructions / s from the wall clock of the launch (hipEvents) -- the figure bench.py's roofline uses:
// it needs no clock assumption -- and (b) wave-instructions per s_memtime tick per SIMD (per wave: median / fastest / slowest).
// s_memtime counters are per XCD and not synchronised with each other (the "eff. GHz" column, ticks over the whole launch / wall time, is
// meaningless across XCDs and only printed for completeness), and the dispatcher does not spread W blocks per CU evenly, so (b) is a
// cross-check of (a) at W = 1 and W = 2 only.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_issue.hip -o tools/microbench/valu_issue
// Run:   tools/microbench/valu_issue [out.json]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#if __has_include("kmix_gen.h")
#include "kmix_gen.h" // the static VALU mix of each path kernel (tools/kernel_mix.py): KMIX_N loop bodies of 96 instructions
#else
#define KMIX_N 0
#endif
enum Op { FMA, PK_FMA, CVT_UBYTE, MIN3, MAX_F, CNDMASK, MUL, ADD_U32, RCP, SQRT, NODE_MIX, FMA_SALU, FMA_MIX, CVT_F16, CVT_U32, BFE_U32, NODE_MIX_F16,
          AND_B32, LSHL_B32, MOV_B32, CMP_F32, ADD_F32, MUL_LO_U32, READLANE, NODE_MIX_SCALAR, N_BASE_OPS, KMIX0 = N_BASE_OPS, N_OPS = N_BASE_OPS + 16 };
static const char *op_names[N_BASE_OPS] = {"v_fma_f32", "v_pk_fma_f32", "v_cvt_f32_ubyte0", "v_min3_f32", "v_max_f32", "v_cndmask_b32", "v_mul_f32",
                                      "v_add_u32", "v_rcp_f32", "v_sqrt_f32", "node_step_mix(24 cvt,12 pk_fma,18 minmax,14 cndmask,12 add/and)", "v_fma_f32 + s_add_u32 (1:1)",
                                      "v_fma_mix_f32 (f16 x f32 + f32)", "v_cvt_f32_f16", "v_cvt_f32_u32", "v_bfe_u32",
                                      "node_step_mix with f16 planes(24 fma_mix,18 minmax,14 cndmask,12 add/and)",
                                      "v_and_b32", "v_lshlrev_b32", "v_mov_b32", "v_cmp_lt_f32", "v_add_f32", "v_mul_lo_u32", "v_readfirstlane_b32",
                                      "node_step_mix_r4(24 cvt,24 fma,18 minmax,14 cndmask,12 add/and)"};

// one instruction on accumulator `a` (and, where it needs them, constants b, c). All streams are independent across the 8 accumulators.
#define I_FMA(a)      asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define I_PKFMA(a)    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(pb), "v"(pc));
#define I_CVT(a)      asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a));
#define I_MIN3(a)     asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
#define I_MAX(a)      asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_CND(a)      asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b));
#define I_MUL(a)      asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_ADDU(a)     asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_RCP(a)      asm volatile("v_rcp_f32 %0, %0" : "+v"(a));
#define I_SQRT(a)     asm volatile("v_sqrt_f32 %0, %0" : "+v"(a));
#define I_FMAMIX(a)   asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(a) : "v"(b), "v"(c));
#define I_CVTH(a)     asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(a));
#define I_CVTU(a)     asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a));
#define I_BFE(a)      asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(a));
#define I_SALU(s)     asm volatile("s_add_u32 %0, %0, 1" : "+s"(s));
#define I_AND(a)      asm volatile("v_and_b32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_LSHL(a)     asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(a));
#define I_MOV(a)      asm volatile("v_mov_b32 %0, %1" : "+v"(a) : "v"(b));
#define I_CMP(a)      asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define I_ADDF(a)     asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_MULLO(a)    asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));
#define I_LANE(a)     asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(s0) : "v"(a));

#define R8(M) M(a0) M(a1) M(a2) M(a3) M(a4) M(a5) M(a6) M(a7)
#define R8P(M) M(p0) M(p1) M(p2) M(p3) M(p4) M(p5) M(p6) M(p7)

template <int OP>
__global__ __launch_bounds__(256) void k_issue(int iters, uint64_t *cycles, float *sink) {
    extern __shared__ float lds[];
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    float b = 0.999f, c = 1e-3f;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0};
    float2v pb = {b, b}, pc = {c, c};
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (threadIdx.x == 1023) lds[0] = a0; // keeps the LDS allocation
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter(); // s_memtime
    for (int it = 0; it < iters; ++it) {
        if (OP == FMA) { R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) R8(I_FMA) }
        if (OP == PK_FMA) { R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) R8P(I_PKFMA) }
        if (OP == CVT_UBYTE) { R8(I_CVT) R8(I_CVT) R8(I_CVT) R8(I_CVT) R8(I_CVT) R8(I_CVT) R8(I_CVT) R8(I_CVT) }
        if (OP == MIN3) { R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) R8(I_MIN3) }
        if (OP == MAX_F) { R8(I_MAX) R8(I_MAX) R8(I_MAX) R8(I_MAX) R8(I_MAX) R8(I_MAX) R8(I_MAX) R8(I_MAX) }
        if (OP == CNDMASK) { R8(I_CND) R8(I_CND) R8(I_CND) R8(I_CND) R8(I_CND) R8(I_CND) R8(I_CND) R8(I_CND) }
        if (OP == MUL) { R8(I_MUL) R8(I_MUL) R8(I_MUL) R8(I_MUL) R8(I_MUL) R8(I_MUL) R8(I_MUL) R8(I_MUL) }
        if (OP == ADD_U32) { R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) R8(I_ADDU) }
        if (OP == RCP) { R8(I_RCP) R8(I_RCP) R8(I_RCP) R8(I_RCP) R8(I_RCP) R8(I_RCP) R8(I_RCP) R8(I_RCP) }
        if (OP == SQRT) { R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) R8(I_SQRT) }
        if (OP == NODE_MIX) { // 80 instructions in the proportions of the BVH4 node step (profiles/r02_notes.md): counted as 80
            R8(I_CVT) R8(I_CVT) R8(I_CVT)                 // 24 v_cvt_f32_ubyteN
            R8P(I_PKFMA) I_PKFMA(p0) I_PKFMA(p1) I_PKFMA(p2) I_PKFMA(p3)   // 12 v_pk_fma_f32
            R8(I_MIN3) R8(I_MAX) I_MIN3(a0) I_MAX(a1)     // 18 min / max
            R8(I_CND) I_CND(a0) I_CND(a1) I_CND(a2) I_CND(a3) I_CND(a4) I_CND(a5)   // 14 v_cndmask
            R8(I_ADDU) I_ADDU(a0) I_ADDU(a1) I_ADDU(a2) I_ADDU(a3)    // 12 integer
        }
        if (OP == FMA_MIX) { R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX) }
        if (OP == CVT_F16) { R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) R8(I_CVTH) }
        if (OP == CVT_U32) { R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) R8(I_CVTU) }
        if (OP == BFE_U32) { R8(I_BFE) R8(I_BFE) R8(I_BFE) R8(I_BFE) R8(I_BFE) R8(I_BFE) R8(I_BFE) R8(I_BFE) }
        if (OP == NODE_MIX_F16) { // the node step if the planes were stored as halves: the conversion rides in the fma (68 instructions)
            R8(I_FMAMIX) R8(I_FMAMIX) R8(I_FMAMIX)
            R8(I_MIN3) R8(I_MAX) I_MIN3(a0) I_MAX(a1)
            R8(I_CND) I_CND(a0) I_CND(a1) I_CND(a2) I_CND(a3) I_CND(a4) I_CND(a5)
            R8(I_ADDU) I_ADDU(a0) I_ADDU(a1) I_ADDU(a2) I_ADDU(a3)
        }
        if (OP == NODE_MIX_SCALAR) { // round 4's node step: the 24 plane distances as scalar fmas instead of 12 v_pk_fma_f32 (92 instructions)
            R8(I_CVT) R8(I_CVT) R8(I_CVT)
            R8(I_FMA) R8(I_FMA) R8(I_FMA)
            R8(I_MIN3) R8(I_MAX) I_MIN3(a0) I_MAX(a1)
            R8(I_CND) I_CND(a0) I_CND(a1) I_CND(a2) I_CND(a3) I_CND(a4) I_CND(a5)
            R8(I_ADDU) I_ADDU(a0) I_ADDU(a1) I_ADDU(a2) I_ADDU(a3)
        }
        if (OP == AND_B32) { R8(I_AND) R8(I_AND) R8(I_AND) R8(I_AND) R8(I_AND) R8(I_AND) R8(I_AND) R8(I_AND) }
        if (OP == LSHL_B32) { R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) R8(I_LSHL) }
        if (OP == MOV_B32) { R8(I_MOV) R8(I_MOV) R8(I_MOV) R8(I_MOV) R8(I_MOV) R8(I_MOV) R8(I_MOV) R8(I_MOV) }
        if (OP == CMP_F32) { R8(I_CMP) R8(I_CMP) R8(I_CMP) R8(I_CMP) R8(I_CMP) R8(I_CMP) R8(I_CMP) R8(I_CMP) }
        if (OP == ADD_F32) { R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) R8(I_ADDF) }
        if (OP == MUL_LO_U32) { R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) R8(I_MULLO) }
        if (OP == READLANE) { R8(I_LANE) R8(I_LANE) R8(I_LANE) R8(I_LANE) R8(I_LANE) R8(I_LANE) R8(I_LANE) R8(I_LANE) }
#if KMIX_N > 0
        if (OP == KMIX0 + 0) { KMIX_BODY_0 }
#endif
#if KMIX_N > 1
        if (OP == KMIX0 + 1) { KMIX_BODY_1 }
#endif
#if KMIX_N > 2
        if (OP == KMIX0 + 2) { KMIX_BODY_2 }
#endif
#if KMIX_N > 3
        if (OP == KMIX0 + 3) { KMIX_BODY_3 }
#endif
#if KMIX_N > 4
        if (OP == KMIX0 + 4) { KMIX_BODY_4 }
#endif
#if KMIX_N > 5
        if (OP == KMIX0 + 5) { KMIX_BODY_5 }
#endif
#if KMIX_N > 6
        if (OP == KMIX0 + 6) { KMIX_BODY_6 }
#endif
#if KMIX_N > 7
        if (OP == KMIX0 + 7) { KMIX_BODY_7 }
#endif
#if KMIX_N > 8
        if (OP == KMIX0 + 8) { KMIX_BODY_8 }
#endif
#if KMIX_N > 9
        if (OP == KMIX0 + 9) { KMIX_BODY_9 }
#endif
        if (OP == FMA_SALU) { // 64 VALU + 64 SALU interleaved: does scalar issue take VALU slots of the same wave / SIMD?
#define FS(a, s) I_FMA(a) I_SALU(s)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
            FS(a0, s0) FS(a1, s1) FS(a2, s2) FS(a3, s3) FS(a4, s0) FS(a5, s1) FS(a6, s2) FS(a7, s3)
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(s0 + s1 + s2 + s3);
    if (r == 1.2345f) sink[0] = r;
    if ((threadIdx.x & 63) == 0) {
        int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        cycles[2 * w] = t0;
        cycles[2 * w + 1] = t1;
    }
}

static int insts_per_iter(int op) { return op >= KMIX0 ? 96 : op == NODE_MIX ? 80 : op == NODE_MIX_F16 ? 68 : op == NODE_MIX_SCALAR ? 92 : 64; }
static const char *name_of(int op) {
#if KMIX_N > 0
    if (op >= KMIX0) return kmix_names[op - KMIX0];
#endif
    return op_names[op];
} // VALU instructions (FMA_SALU: 64 VALU + 64 SALU)

template <int OP>
static void launch(int grid, size_t lds, int iters, uint64_t *cyc, float *sink) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_issue<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_issue<OP>, dim3(grid), dim3(256), lds, 0, iters, cyc, sink);
}

static void launch_op(int op, int grid, size_t lds, int iters, uint64_t *cyc, float *sink) {
    switch (op) {
    case FMA: launch<FMA>(grid, lds, iters, cyc, sink); break;
    case PK_FMA: launch<PK_FMA>(grid, lds, iters, cyc, sink); break;
    case CVT_UBYTE: launch<CVT_UBYTE>(grid, lds, iters, cyc, sink); break;
    case MIN3: launch<MIN3>(grid, lds, iters, cyc, sink); break;
    case MAX_F: launch<MAX_F>(grid, lds, iters, cyc, sink); break;
    case CNDMASK: launch<CNDMASK>(grid, lds, iters, cyc, sink); break;
    case MUL: launch<MUL>(grid, lds, iters, cyc, sink); break;
    case ADD_U32: launch<ADD_U32>(grid, lds, iters, cyc, sink); break;
    case RCP: launch<RCP>(grid, lds, iters, cyc, sink); break;
    case SQRT: launch<SQRT>(grid, lds, iters, cyc, sink); break;
    case NODE_MIX: launch<NODE_MIX>(grid, lds, iters, cyc, sink); break;
    case FMA_SALU: launch<FMA_SALU>(grid, lds, iters, cyc, sink); break;
    case FMA_MIX: launch<FMA_MIX>(grid, lds, iters, cyc, sink); break;
    case CVT_F16: launch<CVT_F16>(grid, lds, iters, cyc, sink); break;
    case CVT_U32: launch<CVT_U32>(grid, lds, iters, cyc, sink); break;
    case BFE_U32: launch<BFE_U32>(grid, lds, iters, cyc, sink); break;
    case NODE_MIX_F16: launch<NODE_MIX_F16>(grid, lds, iters, cyc, sink); break;
    case AND_B32: launch<AND_B32>(grid, lds, iters, cyc, sink); break;
    case LSHL_B32: launch<LSHL_B32>(grid, lds, iters, cyc, sink); break;
    case MOV_B32: launch<MOV_B32>(grid, lds, iters, cyc, sink); break;
    case CMP_F32: launch<CMP_F32>(grid, lds, iters, cyc, sink); break;
    case ADD_F32: launch<ADD_F32>(grid, lds, iters, cyc, sink); break;
    case MUL_LO_U32: launch<MUL_LO_U32>(grid, lds, iters, cyc, sink); break;
    case READLANE: launch<READLANE>(grid, lds, iters, cyc, sink); break;
    case NODE_MIX_SCALAR: launch<NODE_MIX_SCALAR>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 0: launch<KMIX0 + 0>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 1: launch<KMIX0 + 1>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 2: launch<KMIX0 + 2>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 3: launch<KMIX0 + 3>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 4: launch<KMIX0 + 4>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 5: launch<KMIX0 + 5>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 6: launch<KMIX0 + 6>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 7: launch<KMIX0 + 7>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 8: launch<KMIX0 + 8>(grid, lds, iters, cyc, sink); break;
    case KMIX0 + 9: launch<KMIX0 + 9>(grid, lds, iters, cyc, sink); break;
    }
}

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = argc > 2 ? atoi(argv[2]) : 4000;
    const bool quick = argc > 3;
    const size_t lds_total = 160 * 1024;
    uint64_t *cyc;
    float *sink;
    CHECK(hipMalloc(&cyc, (size_t)cus * 8 * 4 * 2 * sizeof(uint64_t)));
    CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::string json = "{\"device\": \"" + std::string(prop.name) + "\", \"gcn_arch\": \"" + std::string(prop.gcnArchName) + "\", \"cus\": " + std::to_string(cus) +
                       ", \"clock_rate_khz_reported\": " + std::to_string(prop.clockRate) + ", \"iters\": " + std::to_string(iters) + ", \"results\": [";
    bool first = true;
    printf("%s (%s), %d CUs, hipDeviceProp clockRate %d kHz\n", prop.name, prop.gcnArchName, cus, prop.clockRate);
    printf("%-60s %5s %12s %12s %12s %10s %10s\n", "instruction stream", "W/SIMD", "inst/clk/SIMD", "(min wave)", "(max wave)", "eff. GHz", "G inst/s");
    for (int op = 0; op < KMIX0 + KMIX_N; ++op) {
        if (op == FMA_SALU && !getenv("VALU_ISSUE_SALU")) continue; // (the per-lane scalar accumulators of this stream compile to readfirstlane loops: off by default)
        for (int w : {1, 2, 3, 4, 6, 8}) {
            if (quick && w != 1 && w != 8) continue; // (under a counter pass every dispatch is slow: two occupancies are enough there)
            const int grid = cus * w;
            const size_t lds = (lds_total / w) & ~(size_t)1023; // W blocks of 256 threads fit one CU, W + 1 do not
            const size_t lds_use = lds > 64 * 1024 ? lds : lds;  // (gfx950 lets one block take all 160 KiB)
            launch_op(op, grid, lds_use - 512, iters / 8, cyc, sink); // warm-up (clocks, code)
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            launch_op(op, grid, lds_use - 512, iters, cyc, sink);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<uint64_t> h((size_t)grid * 4 * 2);
            CHECK(hipMemcpy(h.data(), cyc, h.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
            std::vector<double> span(grid * 4);
            uint64_t lo = ~0ull, hi = 0;
            for (int i = 0; i < grid * 4; ++i) {
                span[i] = (double)(h[2 * i + 1] - h[2 * i]);
                lo = std::min(lo, h[2 * i]);
                hi = std::max(hi, h[2 * i + 1]);
            }
            std::sort(span.begin(), span.end());
            const double n_inst = (double)iters * insts_per_iter(op);
            // a SIMD hosts W waves that run side by side for `span` ticks: it issued W * n_inst wave-instructions in that time
            const double med = span[span.size() / 2], mn = span.front(), mx = span.back();
            const double ipc_med = w * n_inst / med, ipc_fast = w * n_inst / mn, ipc_slow = w * n_inst / mx;
            const double eff_ghz = (double)(hi - lo) / (ms * 1e-3) / 1e9; // s_memtime ticks per wall second over the launch
            const double ginst = (double)grid * 4 * n_inst / (ms * 1e-3) / 1e9;
            printf("%-60s %5d %12.4f %12.4f %12.4f %10.3f %10.1f\n", name_of(op), w, ipc_med, ipc_fast, ipc_slow, eff_ghz, ginst);
            char buf[512];
            snprintf(buf, sizeof buf, "%s{\"op\": \"%s\", \"waves_per_simd\": %d, \"inst_per_clk_per_simd\": %.5f, \"fastest_wave\": %.5f, \"slowest_wave\": %.5f, "
                     "\"memtime_ghz\": %.4f, \"ginst_s_wall\": %.2f, \"launch_ms\": %.5f}", first ? "" : ", ", name_of(op), w, ipc_med, ipc_fast, ipc_slow, eff_ghz, ginst, ms);
            json += buf;
            first = false;
        }
    }
    json += "]}";
    if (argc > 1) {
        FILE *f = fopen(argv[1], "w");
        if (f) { fputs(json.c_str(), f); fputc('\n', f); fclose(f); }
    }
    return 0;
}
