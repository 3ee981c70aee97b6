// hrt_glm.h — restatement of the slice of glm the reference's hot path uses.
//
// The reference links glm (git submodule `dependencies/glm`, URL in
// /root/reference/.gitmodules:4-6, pinned commit unknown, directory empty in
// the snapshot).  This header restates the published semantics of the glm
// functions called on the hot path (SURVEY.md §8 row c lists the call sites):
//   vec3 arithmetic, dot (x+y then +z), cross, length, normalize
//   (v * inversesqrt(dot)), reflect, refract, min/max/clamp (ternary forms),
//   quat(euler) ctor, conjugate, quat*vec3.
//
// It is compiled for BOTH the host (g++, oracle and host plumbing) and the
// device (hipcc, gfx950).  Everything here is built from IEEE-754 basic
// operations (+ - * / sqrt, fmaf) in a fixed order, so a value computed on
// the CPU and on the GPU is bit-identical provided both sides are compiled
// with -ffp-contract=off (they are: see oracle/Makefile and
// __graft_entry__.build()).  That is what lets the parity tests compare the
// HIP path against the CPU oracle per pixel instead of only statistically:
// the reference's self-intersection behaviour (SURVEY Q-2) is a coin flip on
// the last bit of `tScaled` in ITriangle::hit (triangle.cpp:105-109).
//
// The transcendental functions (sin, cos, acos, atan2, log) the reference
// takes from the platform libm are restated as fixed polynomial kernels
// (Cephes single-precision forms) for the same reason: glibc, MSVC's CRT and
// ROCm's ocml all differ in the last ulp, so "the reference's libm" is not a
// single function anyway.  tests/test_glm_math.py bounds their error against
// numpy (<= 4 ulp) on the CPU, and tests/test_gpu_parity.py::test_math_kernels_bit_exact checks CPU == GPU
// bit for bit.
#pragma once

#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define HRT_HD __host__ __device__ inline
#else
#define HRT_HD inline
#endif

namespace hrt {

// ------------------------------------------------------------------ scalars
// glm::min(x,y) = (y < x) ? y : x ; glm::max(x,y) = (x < y) ? y : x
HRT_HD float gmin(float x, float y) { return (y < x) ? y : x; }
HRT_HD float gmax(float x, float y) { return (x < y) ? y : x; }
HRT_HD float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }
// float -> texel index.  glm::clamp lets NaN through ((NaN < lo) and (hi < NaN) are both false), and the reference then
// converts NaN to int -- undefined behaviour (x86: INT_MIN and a wild read; gfx950's v_cvt_i32_f32: 0).  NaN texture
// coordinates do occur (a path that went NaN under quirk Q-4 looks the background up with a NaN direction), so the
// conversion is pinned to the GPU's answer, 0, on every build.
// (fmaxf returns its non-NaN operand; the arguments here are never negative, so nothing else changes.  One v_max_f32 --
// the obvious `x == x ? (int)x : 0` made hipcc spill 592 bytes per lane in k_wf_shade and cost 36 ms per frame.)
HRT_HD int texel_index(float x) { return static_cast<int>(fmaxf(x, 0.0f)); }
HRT_HD double gmind(double x, double y) { return (y < x) ? y : x; }

HRT_HD uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
HRT_HD float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// ------------------------------------------------------------------ vec3
struct vec3 {
    float x, y, z;
    HRT_HD vec3() : x(0.f), y(0.f), z(0.f) {}
    HRT_HD explicit vec3(float s) : x(s), y(s), z(s) {}
    HRT_HD vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    HRT_HD float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
struct vec2 {
    float x, y;
    HRT_HD vec2() : x(0.f), y(0.f) {}
    HRT_HD vec2(float a, float b) : x(a), y(b) {}
};

HRT_HD vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
HRT_HD vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
HRT_HD vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
HRT_HD vec3 operator/(vec3 a, vec3 b) { return vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
HRT_HD vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
HRT_HD vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
HRT_HD vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
HRT_HD vec3 operator+(vec3 a, float s) { return vec3(a.x + s, a.y + s, a.z + s); }
HRT_HD vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
HRT_HD vec3& operator+=(vec3& a, vec3 b) { a = a + b; return a; }
HRT_HD vec3& operator*=(vec3& a, vec3 b) { a = a * b; return a; }
HRT_HD vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }
HRT_HD vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }

// glm::dot(vec3): tmp = a*b; return tmp.x + tmp.y + tmp.z
HRT_HD float dot(vec3 a, vec3 b) {
    float tx = a.x * b.x, ty = a.y * b.y, tz = a.z * b.z;
    return (tx + ty) + tz;
}
HRT_HD vec3 cross(vec3 x, vec3 y) {
    return vec3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
HRT_HD float length(vec3 v) { return sqrtf(dot(v, v)); }
// glm::normalize(v) = v * inversesqrt(dot(v,v)) ; inversesqrt(x) = 1 / sqrt(x)
HRT_HD vec3 normalize(vec3 v) { return v * (1.0f / sqrtf(dot(v, v))); }
// glm::reflect(I,N) = I - N * dot(N,I) * 2
HRT_HD vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
// glm::refract(I,N,eta): k = 1 - eta^2 (1 - dot(N,I)^2); k<0 ? 0 : eta*I - (eta*dot(N,I) + sqrt(k)) * N
HRT_HD vec3 refract(vec3 I, vec3 N, float eta) {
    float d = dot(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return vec3(0.0f);
    return eta * I - (eta * d + sqrtf(k)) * N;
}
// hobbyraytracer.h:34-38 nearZero (1e-8 is a double literal there; |c| is promoted)
HRT_HD bool near_zero(vec3 e) {
    const double s = 1e-8;
    return ((double)fabsf(e.x) < s) && ((double)fabsf(e.y) < s) && ((double)fabsf(e.z) < s);
}

// ------------------------------------------------------------------ quat
struct quat { float x, y, z, w; };
// glm::conjugate
HRT_HD quat conjugate(quat q) { quat r; r.x = -q.x; r.y = -q.y; r.z = -q.z; r.w = q.w; return r; }
// glm: quat * vec3 : uv = cross(q.xyz, v); uuv = cross(q.xyz, uv); v + ((uv * q.w) + uuv) * 2
HRT_HD vec3 rotate(quat q, vec3 v) {
    vec3 qv(q.x, q.y, q.z);
    vec3 uv = cross(qv, v);
    vec3 uuv = cross(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}

// ------------------------------------------------------------------ libm restatement
// Cephes-style single precision kernels, evaluated with plain mul/add in a
// fixed order (no fma contraction on either side).
namespace detail {
HRT_HD float sin_poly(float r) {  // |r| <= pi/4
    float z = r * r;
    float p = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r;
    return p + r;
}
HRT_HD float cos_poly(float r) {  // |r| <= pi/4
    float z = r * r;
    float p = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z;
    return (p - 0.5f * z) + 1.0f;
}
// Cody-Waite reduction by pi/2 in three parts; valid for |x| < 8192 (k has 13 bits, every product is exact inside the fma).
// Every caller but the checker texture has a bounded argument (angles in [0, 2 pi]); that one uses gsin_wide below.
HRT_HD float reduce_pio2(float x, int& q) {
    float k = rintf(x * 0.636619772367581343f);
    q = (int)k;
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188e-8f, r);
    return r;
}
// The same for any argument: beyond 8192 in fp64 with a two-part pi/2 (good to ~1e-9 up to 1e9: the checker texture of
// a floor a few hundred kilometres wide); fp64 is IEEE on both sides, so CPU and GPU still agree bit for bit.  Beyond what
// that can resolve (|x| > 1e15) the answer is pinned to r = 0 (sin 0, cos 1) -- the "total loss of significance" convention
// of the fp32 libm this follows -- and to NaN for inf / NaN, like std::sin.  (Converting k to int directly, as above, is
// undefined above 2^31, and x86 and gfx950 disagree there: a checker evaluated 7e9 units from the origin came out odd on
// one and even on the other; tests/tools/gpu_fuzz.py seed 11081.)
HRT_HD float reduce_pio2_wide(float x, int& q) {
    const float ax = fabsf(x);
    if (ax < 8192.0f) return reduce_pio2(x, q);
    q = 0;
    if (!(ax <= 3.4028234e38f)) return u2f(0x7fc00000u);            // inf, NaN
    if (ax > 1e15f) return 0.0f;
    const double xd = (double)x;
    const double k = rint(xd * 0.63661977236758134308);
    double r = fma(-k, 1.57079632679489655800, xd);                   // pi/2 rounded to double ...
    r = fma(-k, 6.12323399573676603587e-17, r);                       // ... and what the rounding left over
    q = (int)(k - 4.0 * floor(k * 0.25));                             // k mod 4 in [0, 3]
    return (float)r;
}
HRT_HD float asin_core(float a) {  // 0 <= a <= 0.5 : asin(a)
    float z = a * a;
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z +
               1.6666752422e-1f) * z * a;
    return p + a;
}
HRT_HD float atan_core(float x) {  // x >= 0
    float y0;
    if (x > 2.414213562373095f) { y0 = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y0 = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y0 = 0.0f;
    float z = x * x;
    float p = (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return y0 + p;
}
}  // namespace detail

// sin and cos of one angle: one range reduction, each polynomial evaluated once, quadrant handled by
// selects (no per-lane switch: on the GPU a `switch (q & 3)` makes a wave execute all four cases).
// Same operations and results as the textbook form  q&3: 0 -> (s, c)  1 -> (c, -s)  2 -> (-s, -c)  3 -> (-c, s).
HRT_HD void gsincos(float x, float& s_out, float& c_out) {
    int q; float r = detail::reduce_pio2(x, q);
    const float s = detail::sin_poly(r), c = detail::cos_poly(r);
    const bool swap = (q & 1) != 0;
    const float ss = swap ? c : s, cc = swap ? s : c;
    s_out = (q & 2) ? -ss : ss;
    c_out = (((q + 1) & 2) != 0) ? -cc : cc;
}
HRT_HD float gsin(float x) { float s, c; gsincos(x, s, c); return s; }
// sin of an argument of any size (texture.cpp:20-21, the checker: sin(10 * p) of a world-space point)
HRT_HD float gsin_wide(float x) {
    int q; float r = detail::reduce_pio2_wide(x, q);
    const float v = (q & 1) ? detail::cos_poly(r) : detail::sin_poly(r);
    return (q & 2) ? -v : v;
}
HRT_HD float gcos(float x) { float s, c; gsincos(x, s, c); return c; }
HRT_HD float gasin(float x) {
    float a = fabsf(x);
    float r;
    if (a > 1.0f) return u2f(0x7fc00000u);
    if (a > 0.5f) {
        float z = 0.5f * (1.0f - a);
        float s = sqrtf(z);
        r = 1.5707963267948966f - 2.0f * detail::asin_core(s);
    } else {
        r = detail::asin_core(a);
    }
    return x < 0.0f ? -r : r;
}
HRT_HD float gacos(float x) {
    if (x < -1.0f || x > 1.0f) return u2f(0x7fc00000u);
    if (x != x) return x;
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * gasin(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * gasin(sqrtf(0.5f * (1.0f - x)));
    return 1.5707963267948966f - gasin(x);
}
HRT_HD float gatan2(float y, float x) {
    const float PI = 3.14159265358979323846f;
    if (x != x || y != y) return u2f(0x7fc00000u);
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f;
    }
    float a = detail::atan_core(fabsf(y / x));
    if (x < 0.0f) a = PI - a;
    return (y < 0.0f) ? -a : a;
}
HRT_HD float glog(float x) {
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7fc00000u);
    if (x == 0.0f) return u2f(0xff800000u);
    uint32_t u = f2u(x);
    int e = 0;
    if ((u & 0x7f800000u) == 0) {  // subnormal: scale by 2^25
        x = x * 33554432.0f; u = f2u(x); e = -25;
    }
    if ((u & 0x7f800000u) == 0x7f800000u) return x;  // +inf
    e += (int)((u >> 23) & 0xff) - 126;
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);  // [0.5, 1)
    if (m < 0.707106781186547524f) { e -= 1; m = (m + m) - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m +
                    1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m +
               3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + (-2.12194440e-4f * fe);
    y = y + (-0.5f * z);
    float r = m + y;
    r = r + (0.693359375f * fe);
    return r;
}
// (1-c)^5 in double for Dielectric::reflectance (material.h:236-241 calls
// glm::pow(double,int) = std::pow); restated as repeated multiplication.
HRT_HD double pow5(double x) { double x2 = x * x; double x4 = x2 * x2; return x4 * x; }

// glm::quat(eulerAngles) (gtc/quaternion.inl): c = cos(e/2), s = sin(e/2)
HRT_HD quat quat_from_euler(vec3 e) {
    vec3 h = e * 0.5f;
    float cx = gcos(h.x), cy = gcos(h.y), cz = gcos(h.z);
    float sx = gsin(h.x), sy = gsin(h.y), sz = gsin(h.z);
    quat q;
    q.w = cx * cy * cz + sx * sy * sz;
    q.x = sx * cy * cz - cx * sy * sz;
    q.y = cx * sy * cz + sx * cy * sz;
    q.z = cx * cy * sz - sx * sy * cz;
    return q;
}
// glm::radians(deg) = deg * 0.01745329251994329576923690768489
HRT_HD float gradians(float deg) { return deg * 0.01745329251994329576923690768489f; }

}  // namespace hrt
