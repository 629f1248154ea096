// em_policy_math.h - the transcendental pieces of the policy epilogues (mappo_fwd.hip, gmlp.hip), written out so that they
// cost tens of instructions instead of libm's hundreds: the epilogue was half of the policy kernels' time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace {

// Philox4x32-10 (the environment kernels' generator, em_env_math.h, restated for the float32 translation units).  One
// 32x32->64 product per half round (v_mad_u64_u32): as separate v_mul_hi_u32 / v_mul_lo_u32, both quarter-rate, the 40
// multiplies of a block were most of the policy epilogue's issue cycles (five blocks per lane and tile at 65 outputs).
__device__ inline uint4 philox4(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
        const uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// sin and cos of 2*pi*u for u in [0, 1): the argument is reduced EXACTLY (u - j/4 is representable for a 24-bit u), then one
// polynomial pair on [-pi/4, pi/4].  More accurate than sincosf(2*pi*u), whose float argument is already rounded, and
// free of its large-argument path.
__device__ inline void em_sincos2pi(float u, float& s, float& c)
{
#pragma clang fp contract(fast)   // the float32 translation units' mode, also when included where contraction is off (edge_env.hip)
    const float j = rintf(4.0f * u);                       // 0..4
    const float x = 6.2831853071795865f * (u - 0.25f * j); // |x| <= pi/4
    const float z = x * x;
    float ps = 2.7557319224e-6f;                           // 1/9!
    ps = fmaf(ps, z, -1.9841269841e-4f);                   // -1/7!
    ps = fmaf(ps, z, 8.3333333333e-3f);                    // 1/5!
    ps = fmaf(ps, z, -1.6666666667e-1f);                   // -1/3!
    const float sx = fmaf(x * z, ps, x);
    float pc = -2.7557319224e-7f;                          // -1/10!
    pc = fmaf(pc, z, 2.4801587302e-5f);                    // 1/8!
    pc = fmaf(pc, z, -1.3888888889e-3f);                   // -1/6!
    pc = fmaf(pc, z, 4.1666666667e-2f);                    // 1/4!
    pc = fmaf(pc, z, -0.5f);
    const float cx = fmaf(pc, z, 1.0f);
    const int q = (int)j & 3;
    const float s0 = (q & 1) ? cx : sx, c0 = (q & 1) ? sx : cx;
    s = (q & 2) ? -s0 : s0;                                // q: 0 (s, c)  1 (c, -s)  2 (-s, -c)  3 (-c, s)
    c = (q == 1 || q == 2) ? -c0 : c0;
}

// two standard normals from two 32-bit words (Box-Muller, float32); oracle/philox_ref.box_muller restates it
__device__ inline void em_box_muller(uint32_t a, uint32_t b, float& z0, float& z1)
{
#pragma clang fp contract(fast)
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * logf(u1));
    float s, c;
    em_sincos2pi(u2, s, c);
    z0 = rad * c;
    z1 = rad * s;
}

// MAPPOTrainer._format_env_actions, mappo_trainer.py:148-152: float32(0.1 + 0.9 * clip((tanh(float64(raw)) + 1) / 2, 0, 1)).
// tanh(x) = sign(x) (1 - t) / (1 + t) with t = exp(-2|x|) in float64: absolute error a few 1e-16, which is what the float32
// result needs (the relative error near 0, where libm's tanh takes a separate path, does not matter after the + 1).
__device__ inline float em_resource_share(float raw)
{
#pragma clang fp contract(fast)
    const double x = (double)raw;
    const double ax = fabs(x);
    double th = 1.0;
    if (ax < 20.0) {
        const double t = exp(-2.0 * ax);
        th = (1.0 - t) / (1.0 + t);
    }
    if (x != x) th = x;                                    // NaN stays NaN (np.clip passes it through)
    th = x < 0.0 ? -th : th;
    double norm = (th + 1.0) / 2.0;
    norm = norm < 0.0 ? 0.0 : (norm > 1.0 ? 1.0 : norm);
    return (float)(0.1 + 0.9 * norm);
}

}  // namespace
