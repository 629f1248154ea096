// em_actor_tile.h - the shared actor's policy arithmetic for one 16-row tile in accumulator layout, used by the per-step
// rollout kernels (mappo_fwd.hip) and by the whole-episode rollout kernel (edge_env.hip):
//   * policy_sample: Gaussian sample + log-prob (MAPPOActor.get_action, mappo_algorithm.py:62-71,209-213) and the trainer's
//     action formatting (MAPPOTrainer._format_env_actions, mappo_trainer.py:125-158)
//   * actor_tile_global: the 3-layer forward (MAPPOActor.forward, mappo_algorithm.py:36-41) of 16 rows on
//     v_mfma_f32_16x16x4_f32 with the weight fragments read from global memory (L1 hits: the parameters are 23 KB and never
//     change during an episode), in the MFMA order of rollout_policy_kernel, so results are bit-identical to it.
// Every function sets its own floating-point contraction mode (the float32 translation units' default, `fast`), because
// edge_env.hip is compiled with contraction off for the environment's float64 arithmetic.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "edgemarl.h"
#include "em_policy_math.h"

#ifndef FSTAMP
#define FSTAMP(i) do { } while (0)
#endif

namespace {

using f32x4_t = __attribute__((ext_vector_type(4))) float;

enum { P_NOISE = 8 };

struct PolicySampleArgs {
    const float* noise;        // [rows][out_dim] standard normal draws, or nullptr for in-kernel Philox
    int32_t n_nodes, explore;
    uint32_t seed_lo, seed_hi, episode, step, env_offset;
};

// y[r] = mean[row c][o = 4q + r].  All 64 lanes of the wavefront must call it (lane shuffles).  Returns the sampled action
// components act[r] (component 4q + r), and - valid on every lane of the row - the summed log-prob, the placement and the
// resource share.  ls/sd: log_std[4q + r] and its exponential.
__device__ __forceinline__ void policy_sample(const PolicySampleArgs& a, const f32x4_t y, const bool valid, const int c_env,
                                              const int c_agent, const int64_t grow, const int q, const int c, const int O,
                                              const float (&ls_)[4], const float (&sd_)[4], float (&act)[4], float& lp_out,
                                              int& place_out, float& share_out)
{
#pragma clang fp contract(fast)
    const int n = a.n_nodes;
    const int env = valid ? c_env : 0;
    const int agent = valid ? c_agent : 0;
    float lp = 0.0f;
    float best = -__builtin_inff();
    int best_i = 1 << 20;
    bool all_zero = true;
    uint4 rnd = make_uint4(0, 0, 0, 0);
    if (a.explore && !a.noise)
        rnd = philox4(a.env_offset + (uint32_t)env, a.episode, a.step, (P_NOISE << 24) | ((uint32_t)agent << 4) | (uint32_t)q,
                      a.seed_lo, a.seed_hi);
    FSTAMP(9);
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.explore && !a.noise) { em_box_muller(rnd.x, rnd.y, z[0], z[1]); em_box_muller(rnd.z, rnd.w, z[2], z[3]); }
    FSTAMP(10);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 4 * q + r;
        const bool on = o < O;
        float v = y[r];
        if (a.explore && on) {
            const float ls = ls_[r], sd = sd_[r];
            const float e = a.noise ? (valid ? a.noise[grow * O + o] : 0.0f) : z[r];
            v = y[r] + e * sd;                                  // rsample, mappo_algorithm.py:65
            const float d = v - y[r];
            lp += -(d * d) / (2.0f * (sd * sd)) - ls - 0.91893853320467274f;   // Normal.log_prob
        }
        act[r] = v;
        if (on && o < n) {
            if (fabsf(v) > 1e-8f) all_zero = false;
            if (v > best) { best = v; best_i = o; }
        }
    }
    FSTAMP(11);
    // combine the four lanes (q = 0..3) that share a batch row
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        lp += __shfl_xor(lp, m);
        const float ob = __shfl_xor(best, m);
        const int oi = __shfl_xor(best_i, m);
        const int oz = __shfl_xor((int)all_zero, m);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
        all_zero = all_zero && (oz != 0);
    }
    FSTAMP(12);
    // resource share from component n (mappo_trainer.py:148-152): tanh in float64, result float32
    const int src = c + 16 * (n >> 2);
    const float raw = __shfl(act[0], src) * (float)((n & 3) == 0) + __shfl(act[1], src) * (float)((n & 3) == 1) +
                      __shfl(act[2], src) * (float)((n & 3) == 2) + __shfl(act[3], src) * (float)((n & 3) == 3);
    share_out = em_resource_share(raw);
    FSTAMP(13);
    lp_out = a.explore ? lp : 0.0f;                            // eval: log_prob = 0 (mappo_algorithm.py:218)
    place_out = all_zero ? 0 : best_i % n;
}

// Forward of the 14->64->64->11-shaped actor (K1 = n + 4 <= 16 inputs, O = n + 1 <= 16 outputs) for one tile: xb[r] = input
// 4q + r of row c (zero beyond K1).  Weight fragments come from global memory in the layout rollout_policy_kernel stages
// through LDS: layer-1 A operand a1[tt][r] = w1[16 tt + c][4q + r], layer 2 a2[tt][s] = w2[16 tt + c][16 (s >> 2) + 4q + (s & 3)],
// layer 3 a3[s] = w3[c][16 (s >> 2) + 4q + (s & 3)] (rows >= O zero).  Needs w2, w3, b1, b2 on 16-byte boundaries.
__device__ __forceinline__ f32x4_t actor_tile_global(const em_mlp3& net_, const float (&xb)[4], const int q, const int c,
                                                      const int K1, const int O)
{
#pragma clang fp contract(fast)
    constexpr int H = 64;
    f32x4_t h1[4], h2[4], y;
    // The parameter pointers are made opaque per call: inside a long loop (the steps of an episode) the compiler would
    // otherwise hoist every lane's fragment addresses out of the loop and keep - that is, spill - them across the whole body.
    struct { const float *w1, *b1, *w2, *b2, *w3, *b3; } net = {net_.w1, net_.b1, net_.w2, net_.b2, net_.w3, net_.b3};
    asm volatile("" : "+s"(net.w1), "+s"(net.b1), "+s"(net.w2), "+s"(net.b2), "+s"(net.w3), "+s"(net.b3));
    // layer 1
    {
        float a1[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const float* row = net.w1 + (16 * tt + c) * K1;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 4 * q + r;
                const float v = row[k < K1 ? k : K1 - 1];
                a1[tt][r] = k < K1 ? v : 0.0f;
            }
            h1[tt] = *reinterpret_cast<const f32x4_t*>(net.b1 + 16 * tt + 4 * q);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) h1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[tt][s], xb[s], h1[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[tt][r] = fmaxf(h1[tt][r], 0.0f);
    }
    // layer 2: fragments one 16-wide k block (u) at a time, 16 MFMAs per block
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) h2[tt] = *reinterpret_cast<const f32x4_t*>(net.b2 + 16 * tt + 4 * q);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        f32x4_t a2[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) a2[tt] = *reinterpret_cast<const f32x4_t*>(net.w2 + (16 * tt + c) * H + 16 * u + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) h2[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[tt][r], h1[u][r], h2[tt], 0, 0, 0);
    }
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[tt][r] = fmaxf(h2[tt][r], 0.0f);
    // layer 3
    {
        const int cr = c < O ? c : O - 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = 4 * q + r;
            const float v = net.b3[o < O ? o : O - 1];
            y[r] = o < O ? v : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4_t a3 = *reinterpret_cast<const f32x4_t*>(net.w3 + cr * H + 16 * u + 4 * q);
            if (c >= O) a3 = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f32_16x16x4f32(a3[r], h2[u][r], y, 0, 0, 0);
        }
    }
    return y;
}

}  // namespace
