// dmath.h -- scalar-explicit vector arithmetic for the gfx950 kernels.
//
// The kernels evaluate the reference's shading formulas (rendering/**/*.glsl)
// in IEEE binary32 with -ffp-contract=off; fused multiply-adds appear only
// where the reference writes fma() itself (lights/tri.glsl, util.glsl:151-153,
// gltf_bsdf.glsl:237). The GLSL built-ins are pinned to one evaluation order
// (documented in DESIGN.md "Numerics") so that images are reproducible across
// launches, tilings, GPUs counts and against the CPU oracle:
//   dot = (x*x' + y*y') + z*z'   normalize = v * (1/sqrt(dot(v,v)))
//   mix(a,b,t) = a*(1-t) + b*t   reflect(I,N) = I - N*(2*dot(N,I))
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RP_DEV __device__ __forceinline__

// ---- division and square root of the SHADING code (option "fast_math"; k_shade.hip / k_tail.hip are built twice, -DRP_FAST_MATH=0|1).
// 0 (the default): IEEE, correctly rounded -- what the CPU oracle computes statement by statement (~11 VALU instructions per division: two
//    v_div_scale, v_rcp, five fmas, v_div_fmas, v_div_fixup; a third of the glTF shade kernel's 6 900 instructions).
// 1: the hardware's own 1-ulp reciprocal / square root / reciprocal square root (v_rcp_f32, v_sqrt_f32, v_rsq_f32: one quarter-rate
//    instruction each), a / b = a * rcp(b): 4 700 instructions. The GLSL reference's `/`, sqrt() and normalize() are not correctly rounded on
//    a GPU either (Vulkan: 2.5 ulp for division, inversesqrt 2 ulp). Measured (profiles/r06_notes.md section 1): C3 shade launches 1.55 ->
//    1.11 ms, the pipelined C3 frame 3.57 -> 3.34 ms (-6.5 %), C2 -1 %, C4 0. Whole frames against the oracle (tests/test_gpu_whole_frames.py
//    runs both builds): C1 / C2 / C4 / C5 RMSE 3e-6 ... 6e-5, coverage identical, ray counts within 3e-6 -- but C3 2.3e-4 (IEEE: 1.4e-5) with
//    4 290 of 2 M pixels off by more than 1e-3: the GGX lobe of the scene's roughness-0.1 material evaluates 1 + (a^2 - 1) cos^2 with
//    a^2 = 1e-4, which turns an ulp of a normalised half vector into 1e-3 of the lobe (three quarters of the difference go away when only
//    normalize() stays IEEE -- at two thirds of the gain). That is inside north_star's 1e-3, and no further from the truth than the IEEE
//    build (both are an ulp away from the real number), but it is not the oracle's image: small frames at a few samples per pixel exceed 1e-3
//    RMSE on a handful of pixels (tools/soak_fuzz.py). So the IEEE build stays the default -- every parity test runs on what a host that sets
//    nothing gets -- and a host that prefers the 6 % sets the option.
// The traversal (dtraverse.h), the camera ray and the ray offset keep IEEE arithmetic in both builds: ray queries stay bit-exact against
// brute force, primary hits (coverage) identical.
#ifndef RP_FAST_MATH
#define RP_FAST_MATH 0
#endif
#if RP_FAST_MATH
RP_DEV float rp_fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
RP_DEV float rp_frcp(float b) { return __builtin_amdgcn_rcpf(b); }
RP_DEV float rp_fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RP_DEV float rp_frsq(float x) { return __builtin_amdgcn_rsqf(x); }
RP_DEV float rp_fdiv_sqrt(float a, float x) { return a * __builtin_amdgcn_rsqf(x); } // a / sqrt(x)
#else
RP_DEV float rp_fdiv(float a, float b) { return a / b; }
RP_DEV float rp_frcp(float b) { return 1.0f / b; }
RP_DEV float rp_fsqrt(float x) { return sqrtf(x); }
RP_DEV float rp_frsq(float x) { return 1.0f / sqrtf(x); }
RP_DEV float rp_fdiv_sqrt(float a, float x) { return a / sqrtf(x); }
#endif

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};

RP_DEV V2 v2(float x, float y) { return V2{x, y}; }
RP_DEV V3 v3(float x, float y, float z) { return V3{x, y, z}; }
RP_DEV V3 v3s(float s) { return V3{s, s, s}; }

RP_DEV V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RP_DEV V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RP_DEV V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RP_DEV V3 operator/(V3 a, V3 b) { return V3{rp_fdiv(a.x, b.x), rp_fdiv(a.y, b.y), rp_fdiv(a.z, b.z)}; }
RP_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RP_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RP_DEV V3 operator/(V3 a, float s) { return RP_FAST_MATH ? a * rp_frcp(s) : V3{a.x / s, a.y / s, a.z / s}; }
RP_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RP_DEV V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
RP_DEV V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
RP_DEV V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
RP_DEV V2 operator*(V2 a, float s) { return V2{a.x * s, a.y * s}; }
RP_DEV V2 operator/(V2 a, float s) { return RP_FAST_MATH ? a * rp_frcp(s) : V2{a.x / s, a.y / s}; }

RP_DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RP_DEV V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RP_DEV float len3(V3 v) { return rp_fsqrt(dot3(v, v)); }
RP_DEV V3 norm3(V3 v) { return v * rp_frsq(dot3(v, v)); }
RP_DEV float len3_ieee(V3 v) { return sqrtf(dot3(v, v)); } // (the camera ray, the ray offset: the same in both builds)
RP_DEV V3 norm3_ieee(V3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
RP_DEV float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }
RP_DEV V3 mix3(V3 a, V3 b, float t) { return a * (1.0f - t) + b * t; }
RP_DEV float clamp1(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
RP_DEV V3 max3(V3 a, V3 b) { return V3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
RP_DEV V3 abs3(V3 a) { return V3{fabsf(a.x), fabsf(a.y), fabsf(a.z)}; }
RP_DEV V3 reflect3(V3 I, V3 N) { return I - N * (2.0f * dot3(N, I)); }
RP_DEV bool eq3(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

// column-major 3x3
struct M3 {
    V3 c0, c1, c2;
};
RP_DEV V3 mul(const M3 &m, V3 v) {
    return V3{(m.c0.x * v.x + m.c1.x * v.y) + m.c2.x * v.z, (m.c0.y * v.x + m.c1.y * v.y) + m.c2.y * v.z,
              (m.c0.z * v.x + m.c1.z * v.y) + m.c2.z * v.z};
}
// transpose(m) * v
RP_DEV V3 mul_t(const M3 &m, V3 v) { return V3{dot3(m.c0, v), dot3(m.c1, v), dot3(m.c2, v)}; }

RP_DEV float pow2f(float x) { return x * x; }
RP_DEV float pow5f(float x) {
    float x2 = x * x;
    return (x2 * x2) * x;
}
RP_DEV float luminance3(V3 c) { return (0.2126f * c.x + 0.7152f * c.y) + 0.0722f * c.z; }

RP_DEV V3 ld3(const float *p) { return V3{p[0], p[1], p[2]}; }
RP_DEV V3 xyz(float4 v) { return V3{v.x, v.y, v.z}; }
RP_DEV float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
