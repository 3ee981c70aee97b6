// hrt_rng.h — counter-based RNG shared by the HIP kernels, the host plumbing
// and the CPU oracle, plus the restatement of glm's gtc/random distributions.
//
// The reference draws every random number from glm::linearRand /
// sphericalRand / ballRand, i.e. from the unseeded global std::rand()
// (call sites: main.cpp:120-121, material.h:81,118,139,173,218,227,
// constantMedium.cpp:25, bvh.cpp:10).  That
// stream is not reproducible even between two runs of the reference, so
// "seeds pinned" is defined here: Philox4x32-10 (Salmon et al., SC'11) with
//   key     = (seed_lo, seed_hi)
//   counter = (pixel_index, sample_index, bounce, purpose | aux << 8)
// One call yields 4 x u32, enough for any single event of Appendix B.  The key
// contains nothing about tiles, ranks or lanes, so every tiling / GPU count /
// thread schedule produces the identical image.
//
// Documented deviation (Q-11): uniform float = (u32 >> 8) * 2^-24 in [0,1)
// instead of glm's float(u32 built from rand()%255 bytes) / float(UINT32_MAX).
#pragma once
#include "hrt_glm.h"

namespace hrt {

struct u32x4 { uint32_t x, y, z, w; };

enum rng_purpose : uint32_t {
    RNG_JITTER = 0,   // main.cpp:120-121 (bounce field = 0)
    RNG_SCATTER = 1,  // Material::scatter draws of one bounce
    RNG_MEDIUM = 2,   // constantMedium.cpp:25, aux = prim index
    RNG_BALL = 3,     // glm::ballRand rejection loop, aux = attempt
    RNG_BUILD = 4,    // bvh.cpp:10 axis choice (oracle tree build only)
    RNG_LENS = 5      // camera.h:34 glm::circularRand(lensRadius), HRT_FLAG_THIN_LENS only (bounce field = 0)
};

HRT_HD void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0, uint32_t k1) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    uint32_t n0 = hi1 ^ c1 ^ k0;
    uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
}

HRT_HD u32x4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int i = 0; i < 10; ++i) {
        philox_round(c0, c1, c2, c3, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    u32x4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
    return r;
}

// Per-path RNG context: which pixel / sample / bounce is being evaluated.
struct rng_ctx {
    uint32_t seed_lo, seed_hi;
    uint32_t pixel, sample, bounce;
};

HRT_HD u32x4 rng_draw(const rng_ctx& c, uint32_t purpose, uint32_t aux) {
    return philox4x32_10(c.pixel, c.sample, c.bounce, purpose | (aux << 8), c.seed_lo, c.seed_hi);
}

HRT_HD float u01(uint32_t u) { return (float)(u >> 8) * 5.9604644775390625e-8f; }  // 2^-24
HRT_HD double u01d(uint32_t hi, uint32_t lo) {
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (double)(v >> 11) * 1.1102230246251565404e-16;  // 2^-53
}
// glm::linearRand(Min, Max) = u * (Max - Min) + Min
HRT_HD float linear_rand(uint32_t u, float mn, float mx) { return u01(u) * (mx - mn) + mn; }

// glm::sphericalRand(1): theta = linearRand(0, 2pi); phi = acos(linearRand(-1, 1));
// (sin(phi) cos(theta), sin(phi) sin(theta), cos(phi))
HRT_HD vec3 spherical_rand(uint32_t u_theta, uint32_t u_z) {
    float theta = linear_rand(u_theta, 0.0f, 6.283185307179586476925286766559f);
    float phi = gacos(linear_rand(u_z, -1.0f, 1.0f));
    float sp, cp, st, ct;
    gsincos(phi, sp, cp);
    gsincos(theta, st, ct);
    float x = sp * ct;
    float y = sp * st;
    float z = cp;
    return vec3(x, y, z);
}

// glm::circularRand(R): a = linearRand(0, 2 pi); (cos a, sin a) * R   -- a point ON the circle (gtc/random.inl)
HRT_HD void circular_rand(uint32_t u_a, float radius, float& x, float& y) {
    float a = linear_rand(u_a, 0.0f, 6.283185307179586476925286766559f);
    float s, c;
    gsincos(a, s, c);
    x = c * radius; y = s * radius;
}

// glm::ballRand(1): rejection on linearRand(vec3(-1), vec3(1)) until length <= 1.
// One Philox call per attempt (purpose RNG_BALL, aux = attempt).  The loop is
// bounded (P[reject] = 1 - pi/6 per attempt; 64 attempts fail with p ~ 1e-21).
HRT_HD vec3 ball_rand(const rng_ctx& c) {
    vec3 r(0.0f);
    for (uint32_t attempt = 0; attempt < 64; ++attempt) {
        u32x4 u = rng_draw(c, RNG_BALL, attempt);
        r = vec3(linear_rand(u.x, -1.0f, 1.0f), linear_rand(u.y, -1.0f, 1.0f), linear_rand(u.z, -1.0f, 1.0f));
        if (!(length(r) > 1.0f)) break;
    }
    return r;
}

}  // namespace hrt
