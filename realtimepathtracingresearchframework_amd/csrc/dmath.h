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
RP_DEV V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
RP_DEV V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RP_DEV V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RP_DEV V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }
RP_DEV V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RP_DEV V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
RP_DEV V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
RP_DEV V2 operator*(V2 a, V2 b) { return V2{a.x * b.x, a.y * b.y}; }
RP_DEV V2 operator*(V2 a, float s) { return V2{a.x * s, a.y * s}; }
RP_DEV V2 operator/(V2 a, float s) { return V2{a.x / s, a.y / s}; }

RP_DEV float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RP_DEV V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y}; }
RP_DEV float len3(V3 v) { return sqrtf(dot3(v, v)); }
RP_DEV V3 norm3(V3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
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
