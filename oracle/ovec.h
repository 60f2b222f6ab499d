// TEST INFRASTRUCTURE -- part of the CPU oracle, never linked into the product.
//
// ovec.h -- the small subset of GLSL vector arithmetic the reference's shader
// library (rendering/**/*.glsl, dual GLSL/C++ via rendering/language.hpp) relies
// on, written out as plain scalar float operations so that the evaluation order
// is explicit.  Conventions (they are the oracle's definition of the GLSL
// built-ins, which the Vulkan spec leaves implementation-defined in the last
// ulps):
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z          (left to right, no fma)
//   cross(a,b)    = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   length(v)     = sqrtf(dot(v,v))
//   normalize(v)  = v * (1.0f / sqrtf(dot(v,v)))
//   mix(x,y,a)    = x*(1-a) + y*a
//   reflect(I,N)  = I - N*(2*dot(N,I))
//   fma(a,b,c)    = fmaf  (only where the reference writes fma explicitly)
// The translation unit is compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    explicit vec2(float s) : x(s), y(s) {}
    vec2(float x_, float y_) : x(x_), y(y_) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    vec4(vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    float &operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};

static inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
static inline vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
static inline vec2 operator*(vec2 a, vec2 b) { return vec2(a.x * b.x, a.y * b.y); }
static inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
static inline vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
static inline vec2 operator/(vec2 a, float s) { return vec2(a.x / s, a.y / s); }
static inline vec2 operator/(vec2 a, vec2 b) { return vec2(a.x / b.x, a.y / b.y); }
static inline vec2 operator-(vec2 a) { return vec2(-a.x, -a.y); }

static inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
static inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
static inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
static inline vec3 &operator+=(vec3 &a, vec3 b) { a = a + b; return a; }
static inline vec3 &operator-=(vec3 &a, vec3 b) { a = a - b; return a; }
static inline vec3 &operator*=(vec3 &a, vec3 b) { a = a * b; return a; }
static inline vec3 &operator*=(vec3 &a, float s) { a = a * s; return a; }
static inline vec3 &operator/=(vec3 &a, float s) { a = a / s; return a; }
static inline bool all_equal(vec3 a, vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }

static inline vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline vec4 operator-(vec4 a, vec4 b) { return vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline vec4 operator/(vec4 a, float s) { return vec4(a.x / s, a.y / s, a.z / s, a.w / s); }

static inline float dot(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline vec3 cross(vec3 a, vec3 b) {
    return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float length(vec2 v) { return sqrtf(dot(v, v)); }
static inline float length(vec3 v) { return sqrtf(dot(v, v)); }
static inline vec2 normalize(vec2 v) { return v * (1.0f / sqrtf(dot(v, v))); }
static inline vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
static inline float mix(float x, float y, float a) { return x * (1.0f - a) + y * a; }
static inline vec3 mix(vec3 x, vec3 y, float a) { return x * (1.0f - a) + y * a; }
static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
static inline vec3 vmax(vec3 a, vec3 b) { return vec3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static inline vec3 vabs(vec3 a) { return vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline vec3 reflect(vec3 I, vec3 N) { return I - N * (2.0f * dot(N, I)); }

// column-major 3x3 like GLSL mat3(c0,c1,c2)
struct mat3 {
    vec3 c[3];
    mat3() {}
    mat3(vec3 c0, vec3 c1, vec3 c2) { c[0] = c0; c[1] = c1; c[2] = c2; }
    vec3 &operator[](int i) { return c[i]; }
    const vec3 &operator[](int i) const { return c[i]; }
};
static inline vec3 operator*(const mat3 &m, vec3 v) {
    return vec3((m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z,
                (m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z,
                (m.c[0].z * v.x + m.c[1].z * v.y) + m.c[2].z * v.z);
}
static inline mat3 transpose(const mat3 &m) {
    return mat3(vec3(m.c[0].x, m.c[1].x, m.c[2].x), vec3(m.c[0].y, m.c[1].y, m.c[2].y),
                vec3(m.c[0].z, m.c[1].z, m.c[2].z));
}
// GLSL mat3x2 (3 columns of vec2) times vec3
struct mat3x2 {
    vec2 c[3];
};
static inline vec2 operator*(const mat3x2 &m, vec3 v) {
    return vec2((m.c[0].x * v.x + m.c[1].x * v.y) + m.c[2].x * v.z, (m.c[0].y * v.x + m.c[1].y * v.y) + m.c[2].y * v.z);
}
// column-major 2x2
struct mat2 {
    vec2 c[2];
    mat2() {}
    mat2(vec2 c0, vec2 c1) { c[0] = c0; c[1] = c1; }
    explicit mat2(float d) { c[0] = vec2(d, 0); c[1] = vec2(0, d); }
    vec2 &operator[](int i) { return c[i]; }
    const vec2 &operator[](int i) const { return c[i]; }
};
static inline float determinant(const mat2 &m) { return m.c[0].x * m.c[1].y - m.c[1].x * m.c[0].y; }
static inline vec2 operator*(const mat2 &m, vec2 v) {
    return vec2(m.c[0].x * v.x + m.c[1].x * v.y, m.c[0].y * v.x + m.c[1].y * v.y);
}
static inline mat2 operator*(const mat2 &a, const mat2 &b) { return mat2(a * b.c[0], a * b.c[1]); }
static inline mat2 transpose(const mat2 &m) { return mat2(vec2(m.c[0].x, m.c[1].x), vec2(m.c[0].y, m.c[1].y)); }

static inline uint32_t float_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}
static inline float bits_float(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}

} // namespace orc
