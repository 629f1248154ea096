// em_actor_tile.h - the shared actor's policy arithmetic for one 16-row tile in accumulator layout (policy_sample: Gaussian
// sample + log-prob, MAPPOActor.get_action, mappo_algorithm.py:62-71,209-213, and the trainer's action formatting,
// MAPPOTrainer._format_env_actions, mappo_trainer.py:125-158), kept apart from the kernels that call it (mappo_fwd.hip).
// The function sets its own floating-point contraction mode (the float32 translation units' default, `fast`).  That covers
// its own expressions only: the library calls it inlines (logf) still follow the translation unit's -ffp-contract, so a
// unit built with contraction off (edge_env.hip) gets results one ulp away (measured in round 3).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "edgemarl.h"
#include "em_lanes.h"
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
    // (row exchanges on the vector ALU, em_lanes.h: eight ds_bpermute round trips per tile before)
#pragma unroll
    for (int m = 16; m <= 32; m <<= 1) {
        auto other = [&](unsigned v) { return m == 16 ? em_xor16(v, q) : em_xor32(v, q); };
        lp += __uint_as_float(other(__float_as_uint(lp)));
        const float ob = __uint_as_float(other(__float_as_uint(best)));
        const int oi = (int)other((unsigned)best_i);
        const int oz = (int)other((unsigned)all_zero);
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

}  // namespace
