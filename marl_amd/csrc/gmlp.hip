// gmlp.hip - GROUPED 3-layer MLP kernels on the f32 matrix cores for the scaled topology (BASELINE.json configs[4]):
//   * IPPO: one actor (S+4 -> 64 -> 64 -> S+1) and one local critic (S+4 -> 64 -> 64 -> 1) PER DECIDER AGENT, the PPO
//     maths of the reference's MAPPO applied per agent (mappo_algorithm.py:62-71,241-262,293-332; the reference's own
//     IPPO lives in un-vendored ray[rllib], MRRL.PY:611-638: parity unpinned);
//   * MAPPO on the scaled topology: the same kernels with ONE group (shared actor, mappo_algorithm.py:16-71).
// Generalises mlp3_fwd_kernel / mlp3_bwd_kernel (mappo_fwd.hip, mappo_bwd.hip) in two directions: output widths up
// to 80 (OT tiles of 16) and groups of rows that use different networks.
//
// Mapping: a workgroup serves ONE group (network): it stages that network's weights into LDS once and its four
// wavefronts stride over the group's 16-row tiles; `split` workgroups share a group (256 agents x split 1 fills the
// 256 CUs with one agent per CU; a single shared network uses split = 256).  Rows of a group are addressed through a
// permutation (entries stay where the rollout wrote them; only an index is sorted by agent).  The MFMA formulation
// is the transposed problem H^T = W . X^T of mappo_fwd.hip: the accumulator tile of a layer is register for register
// the B operand of the next layer.  Weight gradients accumulate in registers over the group's tiles and leave as one
// slab row per (group, wavefront); reduce_group_slabs sums them in fixed order (bitwise reproducible, no atomics).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "edgemarl.h"
#include "em_common.h"
#include "em_lanes.h"
#include "em_policy_math.h"

#ifndef EM_GBWD_SPLIT_ACTOR
#define EM_GBWD_SPLIT_ACTOR 1
#endif

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int HID = 64;
constexpr int TS = 17;      // row stride of the transpose scratch (floats)
constexpr int S2 = 66;      // LDS row stride of 64-wide weight matrices (2*odd: conflict-free fragment reads)
constexpr float HALF_LOG_2PI = 0.91893853320467274f;

enum { P_NOISE = 8 };
enum { MODE_PLAIN = 0, MODE_POLICY = 1, MODE_VALUE = 2 };

struct GArgs {
    em_gmlp net;
    const float* x;            // [rows][ldx]
    int64_t ldx, n_rows;
    const int64_t* group_off;  // [n_groups + 1] positions in the grouped order, or nullptr (one group = all rows)
    const int32_t* perm;       // [n_rows] grouped position -> row, or nullptr (identity)
    int32_t split, pad0;       // workgroups per group
    // forward outputs
    float* y;                  // MODE_PLAIN: [rows][out_dim]
    const float* log_std;      // [out_dim] of network 0 (same group stride as the network)
    const float* noise;        // [rows][out_dim] standard normal draws or nullptr (in-kernel Philox + Box-Muller)
    float* ent_action;         // [rows][out_dim]
    float* ent_logp;           // [rows]
    const int32_t* ent_env;    // [rows]
    const int32_t* ent_agent;  // [rows]
    const int32_t* ent_dst;    // [rows] index into placement/resource
    int32_t* placement;
    float* resource;
    int32_t n_place, explore;  // the first n_place outputs are placement logits, output n_place is the resource share
    uint32_t seed_lo, seed_hi, episode, step, env_offset, pad1;
    float* value;              // MODE_VALUE: [rows]
    float norm_mean, norm_scale;
    float* x_copy;             // MODE_POLICY, optional: x is read at row ent_dst[row] (the environment's pending lists) and the
                               // row is stored here at [row][ldx] on its way (the entry storage's observation copy)
    // backward
    float* slab;               // [n_groups][split * 4][slab_stride]
    int64_t slab_stride;
    const float* action;       // [rows][out_dim]
    const float* old_logp;     // [rows]
    const float* adv;          // [rows] un-normalised advantages
    const float* ret;          // [rows] returns (critic)
    const double* moments;     // [n_groups][3] {count, sum, sum of squares} of the group's advantages (global)
    float clip, ecoef, vcoef, pad2;
};

// LDS carve-up shared by forward and backward: W1 [64][S1] | W2 [64][S2] | W3 [16*OT][S2] | b1[64] b2[64] b3[16*OT] ls[16*OT]
template <int KU, int OT, bool V4 = false>
struct Lds {
    static constexpr int K1P = KU * 16;
    // V4 (forward-only kernels): row strides of 4 * odd floats, so that a lane's four consecutive fragment values are one
    // aligned ds_read_b128 with a 16-bit immediate offset (no address arithmetic; 16 lanes x 16 B hit 16 different bank
    // quads) instead of two ds_read2_b32 whose 8-bit offsets needed a v_add each
    static constexpr int S1 = V4 ? K1P + 4 : (K1P + 2) | 2;
    static constexpr int S2 = V4 ? HID + 4 : ::S2;
    static constexpr int W1 = 0, W2 = W1 + HID * S1, W3 = W2 + HID * S2, B1 = W3 + 16 * OT * S2, B2 = B1 + HID,
                         B3 = B2 + HID, LS = B3 + 16 * OT, END = LS + 16 * OT;
};

template <int KU, int OT, int NT = 256, bool V4 = false>
__device__ inline void stage_weights(float* lds, const em_gmlp& net, const float* log_std, int g, float* sigma = nullptr,
                                     int K1c = 0, int Oc = 0)
{
    using L = Lds<KU, OT, V4>;
    const int K1 = K1c ? K1c : net.in_dim, O = Oc ? Oc : net.out_dim;
    const int64_t go = (int64_t)g * net.group_stride;
    const float* w1 = net.w1 + go;
    const float* w2 = net.w2 + go;
    const float* w3 = net.w3 + go;
    // the small vectors first: loaded after the matrices they were one more L2 round trip in front of the barrier
    float sb1 = 0.0f, sb2 = 0.0f, sb3 = 0.0f, sls = 0.0f;
    if (threadIdx.x < HID) {
        sb1 = net.b1[go + threadIdx.x];
        sb2 = net.b2[go + threadIdx.x];
    }
    if (threadIdx.x < 16 * OT && (int)threadIdx.x < O) {
        sb3 = net.b3[go + threadIdx.x];
        if (log_std) sls = log_std[go + threadIdx.x];
    }
    // fixed trip counts, read in batches of 8 before anything is written: as plain loops every iteration waited for its own
    // load (56 L2 round trips in a row, 28 k cycles by in-kernel stamps - half of a rollout launch of the policy kernel)
    constexpr int N1 = HID * L::K1P / NT, N2 = HID * HID / NT, N3 = 16 * OT * HID / NT, BATCH = 8;
    static_assert(HID * L::K1P % NT == 0 && 16 * OT * HID % NT == 0 && HID * HID % NT == 0, "whole passes of the workgroup");
#pragma unroll
    for (int j0 = 0; j0 < N1; j0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = (j0 + j) * NT + threadIdx.x, r = i / L::K1P, k = i - r * L::K1P;
            v[j] = (j0 + j < N1 && k < K1) ? w1[r * K1 + k] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = (j0 + j) * NT + threadIdx.x, r = i / L::K1P, k = i - r * L::K1P;
            if (j0 + j < N1) lds[L::W1 + r * L::S1 + k] = v[j];
        }
    }
#pragma unroll
    for (int j0 = 0; j0 < N2; j0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) v[j] = (j0 + j < N2) ? w2[(j0 + j) * NT + threadIdx.x] : 0.0f;
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = (j0 + j) * NT + threadIdx.x;
            if (j0 + j < N2) lds[L::W2 + (i >> 6) * L::S2 + (i & 63)] = v[j];
        }
    }
#pragma unroll
    for (int j0 = 0; j0 < N3; j0 += BATCH) {
        float v[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = (j0 + j) * NT + threadIdx.x;
            v[j] = (j0 + j < N3 && (i >> 6) < O) ? w3[i] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int i = (j0 + j) * NT + threadIdx.x;
            if (j0 + j < N3) lds[L::W3 + (i >> 6) * L::S2 + (i & 63)] = v[j];
        }
    }
    if (threadIdx.x < HID) {
        lds[L::B1 + threadIdx.x] = sb1;
        lds[L::B2 + threadIdx.x] = sb2;
    }
    if (threadIdx.x < 16 * OT) {
        lds[L::B3 + threadIdx.x] = sb3;
        lds[L::LS + threadIdx.x] = sls;
        // sigma = exp(log_std) of every output, once per workgroup (the same expression, so the same bits as evaluating it
        // per element and tile in the epilogue, where it cost 4 x OT exp() per row)
        // behind it: 1 / (2 sigma^2), the log-density's quotient as a product - the expression of the backward kernel's table,
        // so that the rollout's log-probability and the update's first evaluation of it are the same bits
        if (sigma) {
            const float sd = expf(sls);
            sigma[threadIdx.x] = sd;
            sigma[16 * OT + threadIdx.x] = 0.5f * (1.0f / (sd * sd));
        }
    }
}

// forward of one 16-row tile; xb = x fragment (row c, features 16u + 4q + r)
struct NoHook { __device__ void operator()() const {} };
// after_l1 runs once the first layer's products are issued (the x fragment is dead from there: a caller can start loads
// that the epilogue needs into the registers it frees)
// PER_TILE: the weight fragments are re-read from LDS for every tile (the base address passes through an empty asm, so
// the compiler cannot hoist the 400-odd loop-invariant reads out of the caller's tile loop into registers, or, with two
// wavefronts per SIMD and 256 registers, into scratch)
// HEAD1 (value networks, one output): the third layer on the vector ALU - v = b3 + w3 . h2 from the sixteen h2 values the
// lane holds (four partial sums, two lane exchanges), in y[0][0] on every lane of the row - instead of a chain of sixteen
// dependent MFMAs of which 1/16 is used.
// PIPE (backward kernels, two wavefronts per SIMD): the fragments of reduction block u + 1 are requested before the MFMAs of
// block u are issued (a second set of 16 registers), pinned with scheduling fences.
template <int KU, int OT, class Hook = NoHook, bool PER_TILE = false, bool HEAD1 = false, bool PIPE = false>
__device__ inline void forward_tile(const float* lds_in, const float (&xb)[KU * 4], int q, int c, f32x4 (&h1)[4], f32x4 (&h2)[4],
                                    f32x4 (&y)[OT], Hook after_l1 = Hook())
{
    using L = Lds<KU, OT, PER_TILE>;      // the per-tile readers use the ds_read_b128 layout
    const float* lds = lds_in;
    if constexpr (PER_TILE) {
        int zero = 0;
        asm volatile("" : "+v"(zero));
        lds = lds_in + zero;
    }
    auto frag = [&](int base, int u, float (&w)[4]) {   // values k = 16u + 4q + 0..3 of the row at base
        if constexpr (PER_TILE) {
            const float4 v = *reinterpret_cast<const float4*>(lds + base + 16 * u + 4 * q);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = lds[base + 16 * u + 4 * q + r];
        }
    };
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = lds[L::B1 + 16 * t + 4 * q + r];
    if constexpr (PIPE) {
        float w[2][4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) frag(L::W1 + (16 * t + c) * L::S1, 0, w[0][t]);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (u + 1 < KU) {
#pragma unroll
                for (int t = 0; t < 4; ++t) frag(L::W1 + (16 * t + c) * L::S1, u + 1, w[(u + 1) & 1][t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u & 1][t][r], xb[4 * u + r], h1[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
    for (int u = 0; u < KU; ++u)
    {   // reduction index outside, accumulator inside: consecutive MFMAs never wait for each other
        float w[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) frag(L::W1 + (16 * t + c) * L::S1, u, w[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][r], xb[4 * u + r], h1[t], 0, 0, 0);
    }
    }
    after_l1();
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            h1[t][r] = fmaxf(h1[t][r], 0.0f);
            h2[t][r] = lds[L::B2 + 16 * t + 4 * q + r];
        }
    if constexpr (PIPE) {
        float w[2][4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) frag(L::W2 + (16 * t + c) * L::S2, 0, w[0][t]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u + 1 < 4) {
#pragma unroll
                for (int t = 0; t < 4; ++t) frag(L::W2 + (16 * t + c) * L::S2, u + 1, w[(u + 1) & 1][t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int t = 0; t < 4; ++t) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[u & 1][t][r], h1[u][r], h2[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        float w[4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t) frag(L::W2 + (16 * t + c) * L::S2, u, w[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[t][r], h1[u][r], h2[t], 0, 0, 0);
    }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = fmaxf(h2[t][r], 0.0f);
    if constexpr (HEAD1) {
        static_assert(OT == 1, "one output tile");
        float ps[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float w[4];
            frag(L::W3, t, w);                            // row 0 of W3, features 16t + 4q + 0..3
            ps[t] = w[0] * h2[t][0];
#pragma unroll
            for (int r = 1; r < 4; ++r) ps[t] = fmaf(w[r], h2[t][r], ps[t]);
        }
        y[0][0] = em_rows_sum((ps[0] + ps[1]) + (ps[2] + ps[3])) + lds[L::B3];
        y[0][1] = 0.0f; y[0][2] = 0.0f; y[0][3] = 0.0f;
        return;
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[ot][r] = lds[L::B3 + 16 * ot + 4 * q + r];
    // the partial sums of an accumulator are taken in the order k = 0, 1, 2, ... in every variant
#pragma unroll
    for (int u = 0; u < 4; ++u)
    {
        float w[OT][4];
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) frag(L::W3 + (16 * ot + c) * L::S2, u, w[ot]);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) y[ot] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[ot][r], h2[u][r], y[ot], 0, 0, 0);
    }
    // y[ot][r] = out[row c][o = 16 ot + 4q + r]
}

template <int KU, int KC = 0>
__device__ inline void load_x_frag(const GArgs& a, int64_t grow, bool valid, int q, float (&xb)[KU * 4])
{
    const int K1 = KC ? KC : a.net.in_dim;
    const bool vec_ok = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    const float* xr = a.x + (valid ? grow : 0) * a.ldx;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = 16 * u + 4 * q;
        if (vec_ok && k + 3 < K1) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            xb[u * 4 + 0] = v.x; xb[u * 4 + 1] = v.y; xb[u * 4 + 2] = v.z; xb[u * 4 + 3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xb[u * 4 + r] = (k + r) < K1 ? xr[k + r] : 0.0f;
        }
    }
    if (!valid) {
#pragma unroll
        for (int s = 0; s < KU * 4; ++s) xb[s] = 0.0f;
    }
}

// the x fragment back to memory: row grow of a.x_copy (em_gmlp_policy_gather)
template <int KU, int KC = 0>
__device__ inline void store_x_frag(const GArgs& a, int64_t grow, int q, const float (&xb)[KU * 4])
{
    const int K1 = KC ? KC : a.net.in_dim;
    const bool vec_ok = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x_copy) & 15) == 0);
    float* xr = a.x_copy + grow * a.ldx;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = 16 * u + 4 * q;
        if (vec_ok && k + 3 < K1) {
            *reinterpret_cast<float4*>(xr + k) = make_float4(xb[u * 4 + 0], xb[u * 4 + 1], xb[u * 4 + 2], xb[u * 4 + 3]);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k + r < K1) xr[k + r] = xb[u * 4 + r];
        }
    }
}

// NW wavefronts per workgroup (16 by default: 95 vs 101 ms per IPPO rollout with 8, 112 with 4): a group's tiles (about nine per learner and step in the IPPO rollout) are dealt to more
// wavefronts, two per SIMD, so that one's epilogue arithmetic runs under the other's MFMAs
#ifdef EM_STAMP_GFWD   // cycle stamps of every wavefront of one workgroup (tools/stamp_gfwd.py)
__device__ unsigned long long gfwd_stamps[16 * 8];
#define GSTAMP(i) do { if (MODE == MODE_POLICY && blockIdx.x == 7 && (threadIdx.x & 63) == 0) gfwd_stamps[(threadIdx.x >> 6) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define GSTAMP(i) do { } while (0)
#endif
// KC, OC: in_dim and out_dim as compile-time constants (0 = read from the arguments; policy: n_place = OC - 1).  With
// run-time widths every "is this column real" test of the epilogue (3 per element, 20 elements per lane) is a
// loop-invariant lane mask: the compiler hoisted 200 SGPRs of them out of the tile loop, spilled them into VGPR lanes and
// paid a v_readlane plus hazard nops for each one in every tile.
#ifndef EM_GFWD_PIPE
#define EM_GFWD_PIPE (NW > 4)
#endif
template <int KU, int OT, int MODE, int NW = 4, int KC = 0, int OC = 0>
__global__ __launch_bounds__(64 * NW, 1) void gmlp_fwd_kernel(const GArgs a)
{
    GSTAMP(0);
    extern __shared__ __align__(16) float lds[];
    using L = Lds<KU, OT, (NW > 4)>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int O = OC ? OC : a.net.out_dim;
    const int g = blockIdx.x / a.split, part = blockIdx.x - g * a.split;
    const int64_t r0 = a.group_off ? a.group_off[g] : 0, r1 = a.group_off ? a.group_off[g + 1] : a.n_rows;
    const int64_t n_rows = r1 - r0, n_tiles = (n_rows + 15) >> 4;
    if ((int64_t)part * NW >= n_tiles) return;         // uniform over the workgroup: safe before the barrier
    float* sd_t = lds + L::END;
    // A tile's row numbers - the entries of its rows and, for the policy epilogue, their environment, agent and action slot -
    // are fetched by one chain of loads on addresses that are valid for any row (entry 0 for rows past the end).  The first
    // tile's chain is started BEFORE the weights are staged, so that its two dependent memory round trips run beside the
    // staging instead of behind the barrier (a wavefront has one tile per step in the rollout).
    int64_t grow = 0;
    bool valid = false;
    float xb[KU * 4];
    int p_env = 0, p_agent = 0, p_dst = 0;
    auto fetch = [&](int64_t tile_) {
        const int64_t row_ = tile_ * 16 + c;
        valid = tile_ < n_tiles && row_ < n_rows;
        const int64_t at_ = r0 + (valid ? row_ : 0);
        grow = a.perm ? (int64_t)a.perm[at_] : at_;
        if (!valid) grow = 0;
        if constexpr (MODE == MODE_POLICY) {
            p_env = a.ent_env[grow];
            p_agent = a.ent_agent[grow];
            p_dst = a.ent_dst[grow];
        }
    };
    const int64_t tile_stride = (int64_t)a.split * NW;
    fetch((int64_t)part * NW + wave);
    stage_weights<KU, OT, 64 * NW, (NW > 4)>(lds, a.net, MODE == MODE_POLICY ? a.log_std : nullptr, g, MODE == MODE_POLICY ? sd_t : nullptr, KC, OC);
    GSTAMP(1);
    __syncthreads();

    for (int64_t tile = (int64_t)part * NW + wave; tile < n_tiles; tile += tile_stride) {
        GSTAMP(2);
        // (the fragment itself fetched before the staging: 10 spilled registers)
        if constexpr (MODE == MODE_POLICY) {
            const bool gather = a.x_copy != nullptr;     // rows still in the environment's pending lists: read there, copy here
            load_x_frag<KU, KC>(a, gather ? (int64_t)p_dst : grow, valid, q, xb);
            if (gather && valid) store_x_frag<KU, KC>(a, grow, q, xb);
        } else {
            load_x_frag<KU, KC>(a, grow, valid, q, xb);
        }
        f32x4 h1[4], h2[4], y[OT];
        GSTAMP(3);
        forward_tile<KU, OT, NoHook, (NW > 4), (MODE == MODE_VALUE && OT == 1), EM_GFWD_PIPE>(lds, xb, q, c, h1, h2, y);
        GSTAMP(4);

        if constexpr (MODE == MODE_PLAIN) {
            if (valid) {
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * ot + 4 * q + r < O) a.y[grow * O + 16 * ot + 4 * q + r] = y[ot][r];
            }
        } else if constexpr (MODE == MODE_VALUE) {
            if (valid && q == 0) a.value[grow] = y[0][0] * a.norm_scale + a.norm_mean;   // denormalize, mappo_algorithm.py:384-386
        } else {
            // ---- policy epilogue: rsample + log-prob (mappo_algorithm.py:62-71,209-213), action formatting
            //      (mappo_trainer.py:125-158) --------------------------------------------------------------
            const int n = OC ? OC - 1 : a.n_place;
            const int env = valid ? p_env : 0;
            const int agent = valid ? p_agent : 0;
            float lp = 0.0f, best = -__builtin_inff(), res_raw = 0.0f;
            int best_i = 1 << 20;
            bool all_zero = true;
            const float* den_f = sd_t + 16 * OT;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                // the lane's four table entries of this output tile in one 16-byte read each, outside the per-element branches
                const f32x4 ls4 = *reinterpret_cast<const f32x4*>(lds + L::LS + 16 * ot + 4 * q);
                const f32x4 sd4 = *reinterpret_cast<const f32x4*>(sd_t + 16 * ot + 4 * q);
                const f32x4 den4 = *reinterpret_cast<const f32x4*>(den_f + 16 * ot + 4 * q);
                float z[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.explore && !a.noise) {
                    const uint4 rnd = philox4(a.env_offset + (uint32_t)env, a.episode, a.step,
                                              (P_NOISE << 24) | ((uint32_t)agent << 8) | (uint32_t)(ot << 2) | (uint32_t)q,
                                              a.seed_lo, a.seed_hi);
                    em_box_muller(rnd.x, rnd.y, z[0], z[1]);
                    em_box_muller(rnd.z, rnd.w, z[2], z[3]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * ot + 4 * q + r;
                    const bool on = o < O;
                    float v = y[ot][r];
                    if (a.explore && on) {
                        const float e = a.noise ? (valid ? a.noise[grow * O + o] : 0.0f) : z[r];
                        v = y[ot][r] + e * sd4[r];                              // rsample, mappo_algorithm.py:65
                        const float d = v - y[ot][r];
                        lp += -(d * d) * den4[r] - ls4[r] - HALF_LOG_2PI;       // Normal.log_prob (quotient as in the backward kernel)
                    }
                    if (on && valid) a.ent_action[grow * O + o] = v;
                    if (on && o < n) {
                        if (fabsf(v) > 1e-8f) all_zero = false;
                        if (v > best) { best = v; best_i = o; }
                    }
                    if (o == n) res_raw = v;
                }
            }
#pragma unroll
            for (int m = 16; m <= 32; m <<= 1) {   // combine the four lanes (q = 0..3) that share a batch row
                lp += __shfl_xor(lp, m);
                const float ob = __shfl_xor(best, m);
                const int oi = __shfl_xor(best_i, m);
                const int oz = __shfl_xor((int)all_zero, m);
                res_raw += __shfl_xor(res_raw, m);
                if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
                all_zero = all_zero && (oz != 0);
            }
            if (valid && q == 0) {
                a.ent_logp[grow] = a.explore ? lp : 0.0f;                  // eval: log_prob = 0 (mappo_algorithm.py:218)
                const int place = all_zero ? 0 : best_i % n;
                const int64_t dst = p_dst;
                a.placement[dst] = place;
                a.resource[dst] = em_resource_share(res_raw);             // mappo_trainer.py:148-152
            }
            GSTAMP(5);
        }
        if (tile + tile_stride < n_tiles) fetch(tile + tile_stride);
    }
    GSTAMP(6);
}
#ifdef EM_STAMP_GFWD
extern "C" int em_debug_gfwd_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gfwd_stamps), sizeof(gfwd_stamps)); }
#endif

// fused forward-recompute + loss + backward for the rows of one group (mlp3_bwd_kernel generalised)
// STAGE (rows of K1 % 4 == 0 floats, 16-byte aligned): a tile's 16 input rows are copied ONCE, global -> LDS with no register
// in between (global_load_lds_dwordx4), into a per-wavefront double buffer while the previous tile computes; the forward
// takes its x fragments from the copy and the dW1 products read it again (the register version gathered 4 x KU dependent
// scalars per tile from global memory a second time, through the permutation).
// SPLIT (STAGE kernels, 512 threads; mlp3_bwd_kernel in mappo_bwd.hip has the long version): wavefronts w and w + 4 share a
// SIMD and a tile sequence.  Wavefront w runs forward, loss, dZ2/dZ1, dW3 and dW2; wavefront w + 4 stages the pair's input
// tiles one ahead, holds the dW1 accumulators and the layer-1 bias gradient and issues dW1 += dZ1^T X as one dense run of
// register-operand MFMAs per tile.  Three LDS counters per pair order the hand-overs; the sums of every accumulator are taken
// in the order of the one-wavefront kernel.
#ifdef EM_STAMP_GBWD   // cycles per phase of one activation wavefront, summed over its tiles (tools/stamp_gbwd.py)
__device__ unsigned long long gbwd_stamps[2][16];
#define GBSTAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); if ((i) > 0) gb_acc[i] += (unsigned)(n_ - gb_prev); else gb_acc[0] += 1; gb_prev = n_; } while (0)
#else
#define GBSTAMP(i) do { } while (0)
#endif
#ifndef EM_GBWD_PIPE
#define EM_GBWD_PIPE true
#endif
template <int KU, int OT, bool ACTOR, bool STAGE = false, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256, 1) void gmlp_bwd_kernel(const GArgs a)
{
    static_assert(!SPLIT || STAGE, "the gradient wavefront reads the staged input tile");
    extern __shared__ __align__(16) float lds[];
    using L = Lds<KU, OT>;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;   // wave: the pair's index (LDS regions, tiles, slab row)
    const bool grad_wave = SPLIT && (threadIdx.x >> 8) != 0;
    const int q = lane >> 4, c = lane & 15;
    const int K1 = a.net.in_dim, O = a.net.out_dim;
    const int g = blockIdx.x / a.split, part = blockIdx.x - g * a.split;
    const int64_t r0 = a.group_off ? a.group_off[g] : 0, r1 = a.group_off ? a.group_off[g + 1] : a.n_rows;
    const int64_t n_rows = r1 - r0, n_tiles = (n_rows + 15) >> 4;
    float* scr = lds + L::END + wave * (HID * TS);     // per-wavefront transpose scratch [64][TS]
    stage_weights<KU, OT, SPLIT ? 512 : 256>(lds, a.net, ACTOR ? a.log_std : nullptr, g);
    // per-output constants of the Gaussian log-density, evaluated once per workgroup instead of three exp() and a
    // reciprocal per element and tile: 2*sd^2 and 1/sd^2 (same expressions, so the same bits as evaluating them in place)
    float* den_t = lds + L::END + 4 * (HID * TS);
    float* iv_t = den_t + 16 * OT;
    constexpr int NI = (4 * L::K1P + 63) / 64;        // wavefront-wide 16-byte copies per tile (16 rows x K1 floats)
    constexpr int XBUF = NI * 256;                    // floats per staged tile (the tail of the last copy is slack)
    float* xs = iv_t + 16 * OT + wave * (2 * XBUF);
    // SPLIT: {input tiles staged, H2^T/dY^T published, dZ1^T published, dZ1^T read, H2^T/dY^T read} counters of the pair, in
    // tiles, and the pair's hand-over buffer for dW3's operands: H2^T [64][TS] | dY^T [16 OT][TS]
    int* const flg_all = reinterpret_cast<int*>(iv_t + 16 * OT + 4 * (2 * XBUF));
    int* const flg = flg_all + 8 * wave;
    enum { F_XREADY = 0, F_READY3 = 1, F_READY = 2, F_SDONE = 3, F_SDONE3 = 4 };
    constexpr int HB = (HID + 16 * OT) * TS;
    float* const hb = iv_t + 16 * OT + 4 * (2 * XBUF) + 32 + wave * HB;
    if (SPLIT && threadIdx.x < 32) flg_all[threadIdx.x] = 0;
    __syncthreads();
    // hand-over waits: bounded (a wait that runs out poisons the result with NaN and later waits fall through), so that the
    // grid drains whatever happens
    bool poisoned = false;
    auto wait_flag = [&](int idx, int want) {
        if (poisoned) return;
        int spins = 0;
        while (__hip_atomic_load(flg + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { poisoned = true; break; }
        }
    };
    auto post_flag = [&](int idx, int value) {
        if (lane == 0) __hip_atomic_store(flg + idx, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    if constexpr (ACTOR) {
        if (threadIdx.x < 16 * OT) {
            const float sd = expf(lds[L::LS + threadIdx.x]);
            den_t[threadIdx.x] = 0.5f * (1.0f / (sd * sd));   // 1 / (2 sd^2): the log-density's quotient as a product (<= 1 ulp per term)
            iv_t[threadIdx.x] = 1.0f / (sd * sd);
        }
        __syncthreads();
    }

    f32x4 gW1[4][KU], gW2[4][4], gW3[OT][4], gb3[OT], gls[OT];
    float gb1_l = 0.0f, gb2_l = 0.0f;                  // bias gradients of feature `lane`: row sums of the transposed dZ tiles
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < KU; ++u) gW1[t][u] = zero4;
#pragma unroll
        for (int u = 0; u < 4; ++u) gW2[t][u] = zero4;
    }
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
#pragma unroll
        for (int t = 0; t < 4; ++t) gW3[ot][t] = zero4;
        gb3[ot] = zero4; gls[ot] = zero4;
    }
    // loss sums in float64 per lane: a float32 accumulator that adds one row's (nearly constant) entropy a few thousand times
    // rounds the same way every time and drifted 2e-5 relative at 54 M transitions (tests/test_fullsize_gpu.py)
    double loss_sum = 0.0;
    float cnt_sum = 0.0f;

    // the group's own batch statistics: PPO normalises advantages over the learner's batch (mappo_algorithm.py:293)
    const double m_n = a.moments[3 * g], m_s1 = a.moments[3 * g + 1], m_s2 = a.moments[3 * g + 2];
    const float inv_L = m_n > 0.0 ? (float)(1.0 / m_n) : 0.0f;
    float adv_mean = 0.0f, adv_den = 1.0f, ent_row = 0.0f;
    if constexpr (ACTOR) {
        const double mean = m_n > 0.0 ? m_s1 / m_n : 0.0;
        // torch.std (unbiased); a learner with fewer than two transitions gets variance 0 (torch would give NaN)
        double var = m_n > 1.0 ? (m_s2 - m_s1 * m_s1 / m_n) / (m_n - 1.0) : 0.0;
        if (!(var > 0.0)) var = 0.0;
        adv_mean = (float)mean;
        adv_den = (float)sqrt(var) + 1e-8f;
        for (int o = 0; o < O; ++o) ent_row += 0.5f + HALF_LOG_2PI + lds[L::LS + o];            // Normal.entropy summed
    }

    // STAGE: asynchronous copy of a tile, [16][K1] packed.  Lanes past the tile's last valid float4 re-read its first row
    // (always valid) into slack or into rows that every consumer masks.
    // The copy is inline assembly on purpose (see mlp3_bwd_kernel in mappo_bwd.hip): behind the LDS-DMA builtin the compiler
    // waits vmcnt(0) before the next LDS read of any kind; the kernel waits for the copy itself at the end of the tile.
    // grow_t = global row of tile row c (through the permutation), fetched one tile ahead by the caller; lanes 0..15 serve
    // the shuffles, so no index load sits in front of the copy.
    const uint32_t rv = (uint32_t)K1 >> 2, rv_magic = (uint32_t)((0x100000000ull + rv - 1u) / rv);
    auto stage = [&](int64_t tile_, int64_t grow_t, float* dst) {
        const int64_t rows_left = n_rows - tile_ * 16;
        const int nvv = (int)(rows_left < 16 ? rows_left : 16) * (int)rv;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(dst);
        const int g_lo = (int)(uint32_t)grow_t, g_hi = (int)(uint32_t)((uint64_t)grow_t >> 32);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            int idx = lane + 64 * j;
            idx = idx < nvv ? idx : 0;
            const uint32_t rw = __umulhi((uint32_t)idx, rv_magic), v = (uint32_t)idx - rw * rv;
            const int64_t gr = (int64_t)(((uint64_t)(uint32_t)__shfl(g_hi, (int)rw) << 32) | (uint32_t)__shfl(g_lo, (int)rw));
            const float* src = a.x + gr * a.ldx + 4 * v;
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 1024u * (unsigned)j);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(m0v) : "memory", "m0");
        }
    };
    auto global_row = [&](int64_t tile_) -> int64_t {    // global row of (tile_, c)
        const int64_t r_ = tile_ * 16 + c;
        if (tile_ >= n_tiles || r_ >= n_rows) return 0;     // never dereferenced
        return a.perm ? (int64_t)a.perm[r0 + r_] : r0 + r_;
    };
    int cur = 0;
    const int64_t tile_stride = (int64_t)a.split * 4;
    if constexpr (SPLIT) {
        if (grad_wave) {
            float gb1 = 0.0f;
            const int64_t t0 = (int64_t)part * 4 + wave;
            int64_t g_next = global_row(t0 + tile_stride);
            if (t0 < n_tiles) stage(t0, global_row(t0), xs);
            __builtin_amdgcn_s_waitcnt(0);
            post_flag(F_XREADY, 1);
            int k = 0, curg = 0;
            for (int64_t tile = t0; tile < n_tiles; tile += tile_stride, ++k, curg ^= 1) {
                // the other half held tile k - 1: its forward pass is over (F_READY >= k was seen) and this wavefront's copies
                // of its values were consumed by the products
                if (tile + tile_stride < n_tiles) stage(tile + tile_stride, g_next, xs + (curg ^ 1) * XBUF);
                g_next = global_row(tile + 2 * tile_stride);
                const float* xt = xs + curg * XBUF;
                float xv[4][KU];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const bool vb = tile * 16 + 4 * s + q < n_rows;
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const int i = 16 * u + c;
                        xv[s][u] = (vb && i < K1) ? xt[(4 * s + q) * K1 + i] : 0.0f;
                    }
                }
                __builtin_amdgcn_s_waitcnt(0);            // the next tile has landed (and this tile's values are in registers)
                post_flag(F_XREADY, k + 2);
                // dW3[o][j] += sum_b dY^T[o][b] H2^T[j][b], as soon as tile k's loss is known
                if constexpr (ACTOR) {
                wait_flag(F_READY3, k + 1);
                    float tb[4][4];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int s = 0; s < 4; ++s) tb[t][s] = hb[(16 * t + c) * TS + 4 * s + q];
#pragma unroll
                    for (int ot = 0; ot < OT; ++ot) {
                        float ty[4];
#pragma unroll
                        for (int s = 0; s < 4; ++s) ty[s] = hb[(HID + 16 * ot + c) * TS + 4 * s + q];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
#pragma unroll
                            for (int t = 0; t < 4; ++t)
                                gW3[ot][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ty[s], tb[t][s], gW3[ot][t], 0, 0, 0);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    post_flag(F_SDONE3, k + 1);
                }
                wait_flag(F_READY, k + 1);                // the activation wavefront has finished tile k
                float tr[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) tr[t][s] = scr[(16 * t + c) * TS + 4 * s + q];
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; j += 2) { p0 += scr[lane * TS + j]; p1 += scr[lane * TS + j + 1]; }
                gb1 += p0 + p1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                post_flag(F_SDONE, k + 1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < KU; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            gW1[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr[t][s], xv[s][u], gW1[t][u], 0, 0, 0);
            }
            float* out = a.slab + (((int64_t)g * a.split + part) * 4 + wave) * a.slab_stride;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * q + r;
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const int i = 16 * u + c;
                        if (i < K1) out[j * K1 + i] = poisoned ? __builtin_nanf("") : gW1[t][u][r];
                    }
                }
            out[HID * K1 + lane] = gb1;
            const int o_w3g = HID * K1 + HID + HID * HID + HID;
            if constexpr (ACTOR)
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (16 * ot + 4 * q + r < O) out[o_w3g + (16 * ot + 4 * q + r) * HID + 16 * t + c] = gW3[ot][t][r];
            return;
        }
    }
    int kt = 0;                                        // SPLIT: tiles this pair has started
    int64_t grow_cur = global_row((int64_t)part * 4 + wave), grow_next = global_row((int64_t)part * 4 + wave + tile_stride);
    if constexpr (STAGE && !SPLIT) {
        const int64_t t0 = (int64_t)part * 4 + wave;
        if (t0 < n_tiles) stage(t0, grow_cur, xs);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long gb_prev = 0; unsigned gb_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)gb_prev; (void)gb_acc;
    for (int64_t tile = (int64_t)part * 4 + wave; tile < n_tiles; tile += tile_stride) {
        const int64_t row = tile * 16 + c;
        const bool valid = row < n_rows;
        const int64_t grow = valid ? grow_cur : 0;
        GBSTAMP(0);
        const int64_t grow_nn = global_row(tile + 2 * tile_stride);       // needed at the top of the next tile
        const float* xt = xs + cur * XBUF;
        float xb[KU * 4];
        if constexpr (STAGE) {
            if constexpr (SPLIT) wait_flag(F_XREADY, kt + 1);      // staged by the gradient wavefront, one tile ahead
            else if (tile + tile_stride < n_tiles) stage(tile + tile_stride, grow_next, xs + (cur ^ 1) * XBUF);
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int k0 = 16 * u + 4 * q;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid && k0 < K1) xv = *reinterpret_cast<const float4*>(xt + c * K1 + k0);   // K1 % 4 == 0: whole or nothing
                xb[u * 4 + 0] = xv.x; xb[u * 4 + 1] = xv.y; xb[u * 4 + 2] = xv.z; xb[u * 4 + 3] = xv.w;
            }
        } else {
            load_x_frag<KU>(a, grow, valid, q, xb);
        }
        float p_adv = 0.0f, p_old = 0.0f, p_ret = 0.0f;
        if (valid) {
            if constexpr (ACTOR) { p_adv = a.adv[grow]; p_old = a.old_logp[grow]; }
            else p_ret = a.ret[grow];
        }
        f32x4 h1[4], h2[4], y[OT], dy[OT];
        // actor: the row's recorded action, requested once the first layer is issued (into the registers the x fragment
        // frees) and consumed by the loss two layers later; read in place, the loss waited a full memory round trip for it
        float p_act[OT][4];
        auto fetch_actions = [&]() {
            if constexpr (ACTOR) {
#pragma unroll
                for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int o = 16 * ot + 4 * q + r;
                        p_act[ot][r] = (o < O && valid) ? a.action[grow * O + o] : 0.0f;
                    }
            }
        };
        GBSTAMP(1);
        forward_tile<KU, OT, decltype(fetch_actions), false, !ACTOR, EM_GBWD_PIPE>(lds, xb, q, c, h1, h2, y, fetch_actions);

        GBSTAMP(2);
        // ---- loss and dL/dy in accumulator layout: dy[ot][r] = d/d out[row c][o = 16 ot + 4q + r] ------------
        if constexpr (ACTOR) {
            float lp = 0.0f;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                // the lane's four table entries of this output tile in one 16-byte read each, outside the per-element
                // branches (read inside them, every element waited for its own two LDS round trips)
                const f32x4 ls4 = *reinterpret_cast<const f32x4*>(lds + L::LS + 16 * ot + 4 * q);
                const f32x4 den4 = *reinterpret_cast<const f32x4*>(den_t + 16 * ot + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * ot + 4 * q + r;
                    float d = 0.0f;
                    if (o < O && valid) {
                        d = p_act[ot][r] - y[ot][r];
                        lp += -(d * d) * den4[r] - ls4[r] - HALF_LOG_2PI;
                    }
                    dy[ot][r] = d;                                          // holds (action - mu) until g_lp is known
                }
            }
            lp = em_rows_sum(lp);
            float g_lp = 0.0f;
            if (valid) {
                const float A = (p_adv - adv_mean) / adv_den;               // mappo_algorithm.py:293
                const float ratio = __expf(lp - p_old);                     // :306 (|argument| << 1: the hardware exp2 is within 2 ulp)
                const float lo = 1.0f - a.clip, hi = 1.0f + a.clip;
                const float rc = fminf(fmaxf(ratio, lo), hi);
                const float su1 = ratio * A, su2 = rc * A;                  // :307-308
                const float in_range = (ratio >= lo && ratio <= hi) ? 1.0f : 0.0f;
                // d min(s1,s2)/d ratio with autograd's tie rule (equal inputs share the gradient)
                const float wgt = su1 < su2 ? 1.0f : (su1 == su2 ? 0.5f + 0.5f * in_range : in_range);
                g_lp = -(A * ratio * wgt) * inv_L;                          // d(-mean(min))/d logp
                if (q == 0) { loss_sum += (double)fminf(su1, su2); cnt_sum += 1.0f; }
            }
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                const f32x4 iv4 = *reinterpret_cast<const f32x4*>(iv_t + 16 * ot + 4 * q);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * ot + 4 * q + r;
                    const float d = dy[ot][r];
                    float out = 0.0f;
                    if (o < O && valid) {
                        const float iv = iv4[r];
                        out = g_lp * d * iv;                                // d logp / d mu
                        gls[ot][r] += g_lp * (d * d * iv - 1.0f) - a.ecoef * inv_L;   // d logp / d log_std ; entropy bonus
                    }
                    dy[ot][r] = out;
                }
            }
        } else {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) dy[ot] = zero4;
            if (valid && q == 0) {
                const float e = (y[0][0] - a.norm_mean) / a.norm_scale - (p_ret - a.norm_mean) / a.norm_scale;   // :312-314
                loss_sum += (double)(e * e);
                cnt_sum += 1.0f;
                dy[0][0] = a.vcoef * inv_L * e / a.norm_scale;              // d(vcoef * 0.5 * mean(e^2)) / dv
            }
        }

        GBSTAMP(3);
        // (Writing these 36 words under dZ2's MFMAs instead - two per MFMA group - was measured: 23 spilled registers,
        // 31.4 instead of 27.0 ms per epoch.)
        if constexpr (SPLIT && ACTOR) {                // dW3's operands to the gradient wavefront
            wait_flag(F_SDONE3, kt);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[(16 * t + 4 * q + r) * TS + c] = h2[t][r];
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int r = 0; r < 4; ++r) hb[(HID + 16 * ot + 4 * q + r) * TS + c] = dy[ot][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            post_flag(F_READY3, kt + 1);
        }
        GBSTAMP(4);
        // ---- backward through layers 3 and 2 ------------------------------------------------------------------
        f32x4 dz2[4], dz1[4];
        if constexpr (ACTOR) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dz2[t] = zero4;
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) gb3[ot] += dy[ot];
            // dH2^T = W3^T . dY^T, reduction over o = 16 ot + 4q + r.  The four fragments of the NEXT reduction index are
            // requested before the four MFMAs of this one are issued (left to itself the compiler's schedule was "read two
            // words, wait for LDS, two MFMAs", forty times: every LDS latency exposed)
            {
                const float* wcol3 = lds + L::W3 + (4 * q) * S2 + c;
                float fa[4], fb[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) fa[t] = wcol3[16 * t];
#pragma unroll
                for (int i = 0; i < 4 * OT; ++i) {
                    const int ot = i >> 2, r = i & 3;
                    float (&cur_)[4] = (i & 1) ? fb : fa;
                    float (&nxt_)[4] = (i & 1) ? fa : fb;
                    if (i + 1 < 4 * OT) {
                        const int ot1 = (i + 1) >> 2, r1 = (i + 1) & 3;
#pragma unroll
                        for (int t = 0; t < 4; ++t) nxt_[t] = wcol3[(16 * ot1 + r1) * S2 + 16 * t];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) dz2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur_[t], dy[ot][r], dz2[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dz2[t][r] = h2[t][r] > 0.0f ? dz2[t][r] : 0.0f;
                dz1[t] = zero4;
            }
        } else {
            // value head on the vector ALU (see forward_tile<.., HEAD1>): dH2 = w3 * dv through the ReLU, and dW3 accumulates
            // dv * h2 per lane (row c's share; the sixteen rows' lanes are summed when the slab row is written)
            gb3[0] += dy[0];
            const float dv = em_rows_sum(dy[0][0]);       // dL/dv of row c: only the lane with q = 0 holds it, the others 0
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gW3[0][t][r] = fmaf(dv, h2[t][r], gW3[0][t][r]);
                    dz2[t][r] = h2[t][r] > 0.0f ? lds[L::W3 + 16 * t + 4 * q + r] * dv : 0.0f;
                }
                dz1[t] = zero4;
            }
        }
        // HIDE (two-wavefront critic; the actor kernel has no registers left for it: 12 spilled, 28.2 instead of 27.0 ms):
        // everything dW2 needs goes through the scratch UNDER dZ1's MFMAs - dZ2 out, its transposed fragments and the
        // layer-2 bias row sums back, H1 out, its fragments back (80 LDS instructions, five per MFMA group; the LDS runs a
        // wavefront's instructions in order, so one area serves both operands) - instead of two transposes in the open
        constexpr bool HIDE = SPLIT && !ACTOR;
        float tr_a[4][4], tr_b[4][4];
        float pt0 = 0.0f, pt1 = 0.0f;
        auto opT = [&](int i) {
            if (i < 16) scr[(16 * (i >> 2) + 4 * q + (i & 3)) * TS + c] = dz2[i >> 2][i & 3];
            else if (i < 32) tr_a[(i - 16) >> 2][(i - 16) & 3] = scr[(16 * ((i - 16) >> 2) + c) * TS + 4 * ((i - 16) & 3) + q];
            else if (i < 48) { if ((i & 1) == 0) pt0 += scr[lane * TS + (i - 32)]; else pt1 += scr[lane * TS + (i - 32)]; }
            else if (i < 64) scr[(16 * ((i - 48) >> 2) + 4 * q + ((i - 48) & 3)) * TS + c] = h1[(i - 48) >> 2][(i - 48) & 3];
            else tr_b[(i - 64) >> 2][(i - 64) & 3] = scr[(16 * ((i - 64) >> 2) + c) * TS + 4 * ((i - 64) & 3) + q];
        };
        if constexpr (SPLIT) wait_flag(F_SDONE, kt);   // the scratch still holds the previous tile's dZ1^T until it is read
        {   // dH1^T = W2^T . dZ2^T, reduction over k = 16u + 4q + r; fragments one reduction index ahead, as above
            const float* wcol2 = lds + L::W2 + (4 * q) * S2 + c;
            float fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = wcol2[16 * t];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                float (&cur_)[4] = (s & 1) ? fb : fa;
                float (&nxt_)[4] = (s & 1) ? fa : fb;
                if (s + 1 < 16) {
                    const int k1 = 16 * ((s + 1) >> 2) + ((s + 1) & 3);
#pragma unroll
                    for (int t = 0; t < 4; ++t) nxt_[t] = wcol2[k1 * S2 + 16 * t];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) dz1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur_[t], dz2[s >> 2][s & 3], dz1[t], 0, 0, 0);
                if constexpr (HIDE) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) opT(5 * s + i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (HIDE) gb2_l += pt0 + pt1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[t][r] = h1[t][r] > 0.0f ? dz1[t][r] : 0.0f;
        }

        GBSTAMP(5);
        // ---- weight gradients: reduction over the 16 batch rows of the tile ------------------------------------
        // rowsum: += sum over the tile's 16 batch rows of feature `lane` (the bias gradient; rows past n_rows hold zeros)
        auto transpose64 = [&](const f32x4(&m)[4], float(&out)[4][4], float* rowsum) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(16 * t + 4 * q + r) * TS + c] = m[t][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) out[t][s] = scr[(16 * t + c) * TS + 4 * s + q];
            if (rowsum) {
                float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; j += 2) { p0 += scr[lane * TS + j]; p1 += scr[lane * TS + j + 1]; }
                *rowsum += p0 + p1;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        // dW3[o][j] += sum_b dY^T[o][b] H2^T[j][b]
        if constexpr (!SPLIT && ACTOR) {
        transpose64(h2, tr_b, nullptr);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot) {
            float tr_y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(4 * q + r) * TS + c] = dy[ot][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 4; ++s) tr_y[s] = scr[c * TS + 4 * s + q];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    gW3[ot][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_y[s], tr_b[t][s], gW3[ot][t], 0, 0, 0);
        }
        }
        GBSTAMP(6);
        // dW2[j2][j1] += sum_b dZ2^T[j2][b] H1^T[j1][b]
        if constexpr (!HIDE) {
            transpose64(dz2, tr_a, &gb2_l);
            transpose64(h1, tr_b, nullptr);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2) {
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1)
                    gW2[t2][t1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_a[t2][s], tr_b[t1][s], gW2[t2][t1], 0, 0, 0);
                if constexpr (HIDE) {                 // dZ1^T to the scratch under these MFMAs; the gradient wavefront takes it from there
                    const int i = 4 * s + t2;
                    scr[(16 * (i >> 2) + 4 * q + (i & 3)) * TS + c] = dz1[i >> 2][i & 3];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        GBSTAMP(7);
        // dW1[j][i] += sum_b dZ1^T[j][b] X[b][i]
        if constexpr (SPLIT) {
            if constexpr (!HIDE) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) scr[(16 * t + 4 * q + r) * TS + c] = dz1[t][r];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            post_flag(F_READY, kt + 1);
            ++kt;
        } else {
        transpose64(dz1, tr_a, &gb1_l);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rb = tile * 16 + 4 * s + q;
            const bool vb = rb < n_rows;
            const float* xrow;
            if constexpr (STAGE) {
                xrow = xt + (4 * s + q) * K1;
            } else {
                const int64_t gb = vb ? (a.perm ? (int64_t)a.perm[r0 + rb] : r0 + rb) : 0;
                xrow = a.x + gb * a.ldx;
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int i = 16 * u + c;
                const float xv = (vb && i < K1) ? xrow[i] : 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    gW1[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_a[t][s], xv, gW1[t][u], 0, 0, 0);
            }
        }
        }
        if constexpr (STAGE && !SPLIT) {              // the next tile has landed; this tile's half may be overwritten
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (STAGE) cur ^= 1;
        grow_cur = grow_next;
        grow_next = grow_nn;
        GBSTAMP(8);
    }
#ifdef EM_STAMP_GBWD
    if (blockIdx.x == 5 && threadIdx.x == 0)
        for (int i = 0; i < 9; ++i) gbwd_stamps[ACTOR][i] += gb_acc[i];
#endif

    // ---- this wavefront's slab row: [w1 | b1 | w2 | b2 | w3 | b3 | (log_std) | loss, aux, count] ---------------
    float* out = a.slab + (((int64_t)g * a.split + part) * 4 + wave) * a.slab_stride;
    const int o_b1 = HID * K1, o_w2 = o_b1 + HID, o_b2 = o_w2 + HID * HID, o_w3 = o_b2 + HID, o_b3 = o_w3 + O * HID;
    const int o_ls = o_b3 + O, o_loss = ACTOR ? o_ls + O : o_ls;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * q + r;
            if constexpr (!SPLIT) {
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int i = 16 * u + c;
                    if (i < K1) out[j * K1 + i] = gW1[t][u][r];
                }
            }
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1) out[o_w2 + j * HID + 16 * t1 + c] = gW2[t][t1][r];
        }
    if constexpr (!ACTOR) {
        // value head: gW3[0][t][r] is this lane's (row c) share of dW3[0][16t + 4q + r]; sum the 16 rows' lanes (fixed order)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = gW3[0][t][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
                if (c == 0) out[o_w3 + 16 * t + 4 * q + r] = v;
            }
        for (int i = HID + lane; i < O * HID; i += 64) out[o_w3 + i] = 0.0f;    // only output 0 enters the value loss
    } else if constexpr (!SPLIT) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (16 * ot + 4 * q + r < O) out[o_w3 + (16 * ot + 4 * q + r) * HID + 16 * t + c] = gW3[ot][t][r];
    }
    // bias-type gradients: reduce the 16 lanes that share q (fixed butterfly order)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                gb3[ot][r] += __shfl_xor(gb3[ot][r], m);
                gls[ot][r] += __shfl_xor(gls[ot][r], m);
            }
        }
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        loss_sum += __shfl_xor(loss_sum, m);
        cnt_sum += __shfl_xor(cnt_sum, m);
    }
    if constexpr (!SPLIT) out[o_b1 + lane] = gb1_l;
    out[o_b2 + lane] = gb2_l;
    if (poisoned) loss_sum = __builtin_nan("");
    if (c == 0) {
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * ot + 4 * q + r < O) {
                    out[o_b3 + 16 * ot + 4 * q + r] = gb3[ot][r];
                    if constexpr (ACTOR) out[o_ls + 16 * ot + 4 * q + r] = gls[ot][r];
                }
    }
    if (lane == 0) { out[o_loss] = (float)loss_sum; out[o_loss + 1] = (float)((double)ent_row * (double)cnt_sum);   /* the entropy is the same for every row of a group */ out[o_loss + 2] = cnt_sum; }
}

// grads[g][p] = sum over the group's slab rows, fixed order; the last 3 columns (loss sums) go to loss3[g]
__global__ __launch_bounds__(256) void reduce_group_slabs_kernel(const float* __restrict__ slab, int rows_per_group, int64_t stride,
                                                                 float* __restrict__ grads, float* __restrict__ loss3)
{
    const int g = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= stride) return;
    const float* s = slab + (int64_t)g * rows_per_group * stride + p;
    float acc = 0.0f;
    for (int r = 0; r < rows_per_group; ++r) acc += s[(int64_t)r * stride];
    const int64_t np = stride - 3;
    if (p < np) grads[(int64_t)g * np + p] = acc;
    else loss3[(int64_t)g * 3 + (p - np)] = acc;
}

// The same sums for FEW groups with MANY slab rows (a shared network: one group, 1,024 rows of 12.8 k floats): 32 columns x 8
// row lanes per workgroup - row lane j sums the rows r = j (mod 8) in ascending order, the partial sums are combined in the
// order j = 0..7 - so that 400 workgroups share the rows instead of 50 (0.25 ms per epoch).
__global__ __launch_bounds__(256) void reduce_group_slabs_tall_kernel(const float* __restrict__ slab, int rows_per_group, int64_t stride,
                                                                      float* __restrict__ grads, float* __restrict__ loss3)
{
    __shared__ float part[8][32];
    const int g = blockIdx.y, cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int64_t p = (int64_t)blockIdx.x * 32 + cl;
    float acc = 0.0f;
    if (p < stride) {
        const float* s = slab + (int64_t)g * rows_per_group * stride + p;
        int r = rl;
        for (; r + 24 < rows_per_group; r += 32) {
            const float v0 = s[(int64_t)r * stride], v1 = s[(int64_t)(r + 8) * stride], v2 = s[(int64_t)(r + 16) * stride],
                        v3 = s[(int64_t)(r + 24) * stride];
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; r < rows_per_group; r += 8) acc += s[(int64_t)r * stride];
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && p < stride) {
        float t = part[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += part[j][cl];
        const int64_t np = stride - 3;
        if (p < np) grads[(int64_t)g * np + p] = t;
        else loss3[(int64_t)g * 3 + (p - np)] = t;
    }
}

void reduce_group_slabs(const float* slab, int rows_per_group, int64_t stride, int n_groups, float* grads, float* loss3, hipStream_t s)
{
    if (rows_per_group >= 64 && (int64_t)n_groups * ((stride + 255) / 256) < 256) {
        const dim3 grid((unsigned)((stride + 31) / 32), (unsigned)n_groups);
        hipLaunchKernelGGL(reduce_group_slabs_tall_kernel, grid, dim3(256), 0, s, slab, rows_per_group, stride, grads, loss3);
    } else {
        const dim3 grid((unsigned)((stride + 255) / 256), (unsigned)n_groups);
        hipLaunchKernelGGL(reduce_group_slabs_kernel, grid, dim3(256), 0, s, slab, rows_per_group, stride, grads, loss3);
    }
}

// clip_grad_norm_ (mappo_algorithm.py:328-329) + torch.optim.Adam (:331-332), one workgroup per parameter group
struct GroupAdam {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n_per_group;
    float step_size, one_m_beta1, beta2, one_m_beta2, eps, max_norm, bias_c2_sqrt, pad;
};

// `parts` workgroups per group: each one sums the group's squared gradients itself (the same fixed order in every workgroup,
// so all of them clip by the same bits; re-reading 4.5 MB from L2 costs less than a second launch) and then updates its own
// slice.  One workgroup per group took 4.9 ms on the scaled topology's 1.1 M-parameter central critic.
__global__ __launch_bounds__(256) void group_adam_kernel(const GroupAdam a, int parts, float* __restrict__ norms_out)
{
    const int g = blockIdx.x / parts, part = blockIdx.x - g * parts;
    const int64_t base = (int64_t)g * a.n_per_group;
    __shared__ double red[256];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t i = threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(a.grad + base) & 15) == 0) {
        // 16-byte pieces, four of them in flight per thread: every workgroup of a group walks the WHOLE gradient (the 1.1 M
        // parameters of the central critic: 1,104 dependent batches of four 4-byte loads took 0.34 ms per step)
        const float4* g4 = reinterpret_cast<const float4*>(a.grad + base);
        const int64_t n4 = a.n_per_group >> 2;
        auto sq = [](const float4 v) { return ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w); };
        int64_t j = threadIdx.x;
        for (; j + 768 < n4; j += 1024) {
            const float4 v0 = g4[j], v1 = g4[j + 256], v2 = g4[j + 512], v3 = g4[j + 768];
            s0 += sq(v0); s1 += sq(v1); s2 += sq(v2); s3 += sq(v3);
        }
        for (; j < n4; j += 256) s0 += sq(g4[j]);
        i = 4 * n4 + threadIdx.x;
    } else {
        for (; i + 768 < a.n_per_group; i += 1024) {
            const double g0 = (double)a.grad[base + i], g1 = (double)a.grad[base + i + 256], g2 = (double)a.grad[base + i + 512],
                         g3 = (double)a.grad[base + i + 768];
            s0 += g0 * g0; s1 += g1 * g1; s2 += g2 * g2; s3 += g3 * g3;
        }
    }
    for (; i < a.n_per_group; i += 256) s0 += (double)a.grad[base + i] * (double)a.grad[base + i];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const float total_norm = (float)sqrt(red[0]);
    float coef = a.max_norm / (total_norm + 1e-6f);
    coef = coef > 1.0f ? 1.0f : coef;
    if (threadIdx.x == 0 && part == 0 && norms_out) norms_out[g] = total_norm;
    if (!(total_norm <= 3.0e38f)) return;        // as adam_kernel: a non-finite gradient never reaches the parameters
    const int64_t per = (a.n_per_group + parts - 1) / parts;
    const int64_t i0 = (int64_t)part * per, i1 = i0 + per < a.n_per_group ? i0 + per : a.n_per_group;
    for (int64_t k = i0 + threadIdx.x; k < i1; k += 256) {
        const float gr = a.grad[base + k] * coef;
        const float m = a.exp_avg[base + k] + (gr - a.exp_avg[base + k]) * a.one_m_beta1;
        const float v = a.exp_avg_sq[base + k] * a.beta2 + a.one_m_beta2 * gr * gr;
        a.exp_avg[base + k] = m;
        a.exp_avg_sq[base + k] = v;
        const float denom = sqrtf(v) / a.bias_c2_sqrt + a.eps;
        a.param[base + k] = a.param[base + k] - a.step_size * (m / denom);
    }
}

// {count, sum, sum of squares} of the float32 advantages of each group, fixed order (one workgroup per group)
__global__ __launch_bounds__(256) void group_moments_kernel(const float* __restrict__ adv, const int64_t* __restrict__ group_off,
                                                            const int32_t* __restrict__ perm, double* __restrict__ moments)
{
    const int g = blockIdx.x;
    const int64_t r0 = group_off[g], r1 = group_off[g + 1];
    __shared__ double red[2][256];
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = r0 + threadIdx.x; i < r1; i += 256) {
        const double v = (double)adv[perm ? perm[i] : i];
        s1 += v;
        s2 += v * v;
    }
    red[0][threadIdx.x] = s1; red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { red[0][threadIdx.x] += red[0][threadIdx.x + off]; red[1][threadIdx.x] += red[1][threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { moments[3 * g] = (double)(r1 - r0); moments[3 * g + 1] = red[0][0]; moments[3 * g + 2] = red[1][0]; }
}

// GAE of independent learners: one chain per (environment, agent) over that agent's own transitions in time order
// (MAPPO.compute_gae's recursion, mappo_algorithm.py:250-258, with the agent's own value estimates; next_value = 0
// closes every chain, :284).  One lane per environment walks its steps backwards; chain state lives in `state`.
#ifndef EM_GAE_SUB
#define EM_GAE_SUB 16
#endif
constexpr int GAE_SUB = EM_GAE_SUB;
// LDS_STATE: the chain states of the workgroup's 256 / GAE_SUB environments live in LDS (16 D bytes each).  In global memory
// every step's dependent state load sits BEHIND the look-ahead loads of the following steps in the in-order vmcnt queue, so
// waiting for it waited for them as well and each of the 1000 steps cost a whole memory round trip, whatever was prefetched.
template <bool LDS_STATE>
__global__ __launch_bounds__(256) void gae_agents_kernel(int64_t B, int T, int D, const double* __restrict__ reward_tb,
                                                         const uint8_t* __restrict__ done_tb, const uint8_t* __restrict__ k_tb,
                                                         const int32_t* __restrict__ first_tb, const int32_t* __restrict__ ent_agent,
                                                         const float* __restrict__ ent_value, double gamma, double lam,
                                                         double* __restrict__ state, float* __restrict__ adv, float* __restrict__ ret)
{
#pragma clang fp contract(off)   // the float64 recursion in the reference's operation order, no fused multiply-add
    // GAE_SUB lanes per environment: lane `sub` owns the chains of the agents with d % GAE_SUB == sub (chains of different
    // agents never meet), which gives the launch GAE_SUB times the wavefronts to hide the chain-state round trips with
    // (16,384 environments are a quarter of a wavefront per SIMD at one lane each)
    const int64_t gt = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t b = gt / GAE_SUB;
    const int sub = (int)(gt - b * GAE_SUB);
    if (b >= B) return;
    extern __shared__ __align__(16) double st_lds[];
    double* st = LDS_STATE ? st_lds + (threadIdx.x / GAE_SUB) * 2 * (int64_t)D : state + b * 2 * (int64_t)D;   // {gae, v_next} per agent
    for (int d = sub; d < D; d += GAE_SUB) { st[2 * d] = 0.0; st[2 * d + 1] = 0.0; }
    // Four dependent memory round trips per step (k -> first -> entry -> chain state) and one lane has nothing else to do
    // while it waits: the next step's {k, first, reward, done} and its first KP entries {agent, value} are requested one step
    // ahead, which leaves the chain-state access as the only exposed one (6.0 -> 2 ms at 16,384 environments x 1000 steps).
    constexpr int KP = 4;
    struct StepIn { int k; int64_t first; double r, nd; int d[KP]; float v[KP]; };
    // every look-ahead load is unconditional on a clamped address (a load under a per-lane branch makes the compiler wait for
    // everything in flight where the branch ends); what it fetched beyond the step's k entries is never used
    auto fetch = [&](int t, StepIn& o) {
        const int64_t i = (int64_t)(t < 0 ? 0 : t) * B + b;
        const int kk = k_tb[i];
        o.k = t < 0 ? 0 : kk;
        o.r = reward_tb[i];
        o.nd = 1.0 - (double)done_tb[i];
        o.first = first_tb[i];
    };
    auto fetch_entries = [&](StepIn& o) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int64_t e = j < o.k ? o.first + j : 0;
            o.d[j] = ent_agent[e]; o.v[j] = ent_value[e];
        }
    };
    StepIn cur, nxt, nn;
    fetch(T - 1, cur);
    fetch_entries(cur);
    fetch(T - 2, nxt);
    for (int t = T - 1; t >= 0; --t) {
        fetch_entries(nxt);                               // entries of step t - 1 (its {k, first} arrived during step t + 1)
        fetch(t - 2, nn);
        const int k = cur.k;
        const double r = cur.r, nd = cur.nd;
        const int64_t first = cur.first;
        for (int j = 0; j < k; ++j) {
            int d;
            double v;
            if (j < KP) {
                d = cur.d[0]; v = (double)cur.v[0];
#pragma unroll
                for (int q = 1; q < KP; ++q)
                    if (j == q) { d = cur.d[q]; v = (double)cur.v[q]; }
            } else {
                d = ent_agent[first + j];
                v = (double)ent_value[first + j];
            }
            if (d % GAE_SUB != sub) continue;
            const double delta = r + gamma * st[2 * d + 1] * nd - v;
            const double gae = delta + gamma * lam * nd * st[2 * d];
            st[2 * d] = gae;
            st[2 * d + 1] = v;
            adv[first + j] = (float)gae;
            ret[first + j] = (float)(gae + v);
        }
        cur = nxt;
        nxt = nn;
    }
}

int check_gnet(const em_gmlp* net)
{
    if (!net || !net->w1 || !net->b1 || !net->w2 || !net->b2 || !net->w3 || !net->b3)
        return em::fail(EM_ERR_INVALID, "gmlp: null parameter pointer");
    if (net->hidden != HID || net->in_dim < 1 || net->in_dim > 80 || net->out_dim < 1 || net->out_dim > 80)
        return em::fail(EM_ERR_INVALID, "gmlp: unsupported shape (hidden 64, in_dim <= 80, out_dim <= 80)");
    if (net->n_groups < 1 || net->n_groups > 65535) return em::fail(EM_ERR_INVALID, "gmlp: n_groups must be 1..65535");
    return EM_OK;
}

size_t lds_bytes(int ku, int ot, bool bwd, bool stage = false, bool v4 = false)
{
    const int s1 = v4 ? ku * 16 + 4 : (ku * 16 + 2) | 2, s2 = v4 ? HID + 4 : S2;
    // forward: + sigma and 1 / (2 sigma^2) tables of the policy epilogue (32 ot); backward: + transpose scratch and the two loss tables
    return sizeof(float) * ((size_t)HID * s1 + (size_t)HID * s2 + 16 * ot * s2 + 2 * HID + 32 * ot + (bwd ? 4 * HID * TS + 32 * ot : 32 * ot) +
                            (stage ? 4 * 2 * (size_t)((4 * ku * 16 + 63) / 64) * 256 : 0));
}

template <typename K>
int set_lds(K kernel, size_t lds)
{
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return em::hip_fail(e, "hipFuncSetAttribute");
    }
    return EM_OK;
}

#ifndef EM_GFWD_WAVES
#define EM_GFWD_WAVES 16
#endif
// in_dim <= 80 -> KU = 5 (one instantiation keeps the library small; narrower inputs are zero padded in LDS)
template <int MODE>
int launch_gfwd(const GArgs& ga, hipStream_t s)
{
    const int ot = (ga.net.out_dim + 15) / 16;
    constexpr int NW = EM_GFWD_WAVES;
    const dim3 grid((unsigned)(ga.net.n_groups * ga.split)), block(64 * NW);
#define EM_GF(OT_)                                                                                    \
    case OT_: {                                                                                       \
        const size_t lds = lds_bytes(5, OT_, false, false, NW > 4);                                   \
        const int rc = set_lds(gmlp_fwd_kernel<5, OT_, MODE, NW>, lds);                               \
        if (rc) return rc;                                                                            \
        hipLaunchKernelGGL((gmlp_fwd_kernel<5, OT_, MODE, NW>), grid, block, lds, s, ga);             \
    } break
    if constexpr (MODE == MODE_POLICY) {
        // the scaled topology's shape (64 servers: 68 inputs, 64 placement logits + the resource share)
        if (ga.net.in_dim == 68 && ga.net.out_dim == 65 && ga.n_place == 64) {
            const size_t lds = lds_bytes(5, 5, false, false, NW > 4);
            const int rc = set_lds(gmlp_fwd_kernel<5, 5, MODE, NW, 68, 65>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((gmlp_fwd_kernel<5, 5, MODE, NW, 68, 65>), grid, block, lds, s, ga);
            return em::hip_check(hipGetLastError(), "gmlp_fwd_kernel launch");
        }
    }
    switch (ot) {
        EM_GF(1); EM_GF(2); EM_GF(3); EM_GF(4); EM_GF(5);
        default: return em::fail(EM_ERR_INVALID, "gmlp: unsupported out_dim");
    }
#undef EM_GF
    return em::hip_check(hipGetLastError(), "gmlp_fwd_kernel launch");
}

template <bool ACTOR>
int launch_gbwd(const GArgs& ga, hipStream_t s)
{
    const int ot = (ga.net.out_dim + 15) / 16;
    // staged input tiles (see gmlp_bwd_kernel): rows of a multiple of four floats on 16-byte boundaries
    const bool stage = (ga.net.in_dim & 3) == 0 && (ga.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(ga.x) & 15) == 0;
    // two wavefronts per SIMD with the dW1 accumulators in the second one (SPLIT); EM_BWD_SPLIT=0 keeps the one-wavefront
    // kernel (tests compare the two bit for bit)
    const char* split_env = getenv("EM_BWD_SPLIT");
    const bool split2 = stage && (!ACTOR || EM_GBWD_SPLIT_ACTOR) && !(split_env && split_env[0] == '0');
    const dim3 grid((unsigned)(ga.net.n_groups * ga.split)), block(split2 ? 512 : 256);
#define EM_GB_ONE(OT_, STAGE_, SPLIT_)                                                                \
    {                                                                                                 \
        const size_t lds = lds_bytes(5, OT_, true, STAGE_) + (SPLIT_ ? 128 + 4 * sizeof(float) * (HID + 16 * OT_) * TS : 0); \
        const int rc = set_lds(gmlp_bwd_kernel<5, OT_, ACTOR, STAGE_, SPLIT_>, lds);                  \
        if (rc) return rc;                                                                            \
        hipLaunchKernelGGL((gmlp_bwd_kernel<5, OT_, ACTOR, STAGE_, SPLIT_>), grid, block, lds, s, ga); \
    }
#define EM_GB(OT_)                                                                                    \
    case OT_: {                                                                                       \
        if (split2) EM_GB_ONE(OT_, true, true)                                                        \
        else if (stage) EM_GB_ONE(OT_, true, false)                                                   \
        else EM_GB_ONE(OT_, false, false)                                                             \
    } break
    if constexpr (ACTOR) {
        switch (ot) {
            EM_GB(1); EM_GB(2); EM_GB(3); EM_GB(4); EM_GB(5);
            default: return em::fail(EM_ERR_INVALID, "gmlp: unsupported out_dim");
        }
    } else {
        switch (ot) {
            EM_GB(1);
            default: return em::fail(EM_ERR_INVALID, "gmlp critic: out_dim must be 1");
        }
    }
#undef EM_GB
#undef EM_GB_ONE
    return em::hip_check(hipGetLastError(), "gmlp_bwd_kernel launch");
}

int check_rows(const em_gmlp* net, const float* x, int64_t ldx, int64_t n_rows, const int64_t* group_off, int split)
{
    if (!x || n_rows < 0 || ldx < net->in_dim) return em::fail(EM_ERR_INVALID, "gmlp: bad input rows");
    if (net->n_groups > 1 && !group_off) return em::fail(EM_ERR_INVALID, "gmlp: group_off is required with several groups");
    if (split < 1 || (int64_t)split * net->n_groups > (1 << 20)) return em::fail(EM_ERR_INVALID, "gmlp: bad split");
    return EM_OK;
}

void fill_common(GArgs& ga, const em_gmlp* net, const float* x, int64_t ldx, int64_t n_rows, const int64_t* group_off,
                 const int32_t* perm, int split)
{
    std::memset(&ga, 0, sizeof ga);
    ga.net = *net; ga.x = x; ga.ldx = ldx; ga.n_rows = n_rows; ga.group_off = group_off; ga.perm = perm; ga.split = split;
}

}  // namespace

extern "C" {
#ifdef EM_STAMP_GBWD
int em_debug_gbwd_stamps(unsigned long long* out, int reset)
{
    if (reset) { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(gbwd_stamps), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gbwd_stamps), sizeof(unsigned long long) * 32);
}
#endif

int em_gmlp_forward(const em_gmlp* net, const float* x, int64_t ldx, int64_t n_rows, const int64_t* group_off,
                    const int32_t* perm, int32_t split, float* y, void* stream)
{
    int rc = check_gnet(net);
    if (rc) return rc;
    rc = check_rows(net, x, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!y) return em::fail(EM_ERR_INVALID, "em_gmlp_forward: null output");
    if (n_rows == 0) return EM_OK;
    GArgs ga;
    fill_common(ga, net, x, ldx, n_rows, group_off, perm, split);
    ga.y = y;
    return launch_gfwd<MODE_PLAIN>(ga, (hipStream_t)stream);
}

int em_gmlp_policy(const em_gmlp* actor, const float* log_std, const float* x, int64_t ldx, int64_t n_rows,
                   const int64_t* group_off, const int32_t* perm, int32_t split, const float* noise, int explore, uint64_t seed,
                   uint32_t episode, uint32_t rng_step, uint64_t env_offset, const int32_t* ent_env, const int32_t* ent_agent,
                   const int32_t* ent_dst, int32_t n_place, float* ent_action, float* ent_logp, int32_t* placement,
                   float* resource, void* stream)
{
    int rc = check_gnet(actor);
    if (rc) return rc;
    rc = check_rows(actor, x, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!log_std || !ent_env || !ent_agent || !ent_dst || !ent_action || !ent_logp || !placement || !resource)
        return em::fail(EM_ERR_INVALID, "em_gmlp_policy: null argument");
    if (n_place < 1 || n_place + 1 != actor->out_dim) return em::fail(EM_ERR_INVALID, "em_gmlp_policy: out_dim must be n_place + 1");
    if (n_rows == 0) return EM_OK;
    GArgs ga;
    fill_common(ga, actor, x, ldx, n_rows, group_off, perm, split);
    ga.log_std = log_std; ga.noise = noise; ga.ent_action = ent_action; ga.ent_logp = ent_logp; ga.ent_env = ent_env;
    ga.ent_agent = ent_agent; ga.ent_dst = ent_dst; ga.placement = placement; ga.resource = resource; ga.n_place = n_place;
    ga.explore = explore ? 1 : 0;
    ga.seed_lo = (uint32_t)seed; ga.seed_hi = (uint32_t)(seed >> 32); ga.episode = episode; ga.step = rng_step;
    ga.env_offset = (uint32_t)env_offset;
    return launch_gfwd<MODE_POLICY>(ga, (hipStream_t)stream);
}

int em_gmlp_policy_gather(const em_gmlp* actor, const float* log_std, const float* x_src, float* x_copy, int64_t ldx, int64_t n_rows,
                          const int64_t* group_off, const int32_t* perm, int32_t split, const float* noise, int explore,
                          uint64_t seed, uint32_t episode, uint32_t rng_step, uint64_t env_offset, const int32_t* ent_env,
                          const int32_t* ent_agent, const int32_t* ent_dst, int32_t n_place, float* ent_action, float* ent_logp,
                          int32_t* placement, float* resource, void* stream)
{
    int rc = check_gnet(actor);
    if (rc) return rc;
    rc = check_rows(actor, x_src, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!x_copy || !log_std || !ent_env || !ent_agent || !ent_dst || !ent_action || !ent_logp || !placement || !resource)
        return em::fail(EM_ERR_INVALID, "em_gmlp_policy_gather: null argument");
    if (n_place < 1 || n_place + 1 != actor->out_dim) return em::fail(EM_ERR_INVALID, "em_gmlp_policy_gather: out_dim must be n_place + 1");
    if (n_rows == 0) return EM_OK;
    GArgs ga;
    fill_common(ga, actor, x_src, ldx, n_rows, group_off, perm, split);
    ga.x_copy = x_copy;
    ga.log_std = log_std; ga.noise = noise; ga.ent_action = ent_action; ga.ent_logp = ent_logp; ga.ent_env = ent_env;
    ga.ent_agent = ent_agent; ga.ent_dst = ent_dst; ga.placement = placement; ga.resource = resource; ga.n_place = n_place;
    ga.explore = explore ? 1 : 0;
    ga.seed_lo = (uint32_t)seed; ga.seed_hi = (uint32_t)(seed >> 32); ga.episode = episode; ga.step = rng_step;
    ga.env_offset = (uint32_t)env_offset;
    return launch_gfwd<MODE_POLICY>(ga, (hipStream_t)stream);
}

int em_gmlp_value(const em_gmlp* critic, const float* x, int64_t ldx, int64_t n_rows, const int64_t* group_off,
                  const int32_t* perm, int32_t split, float value_norm_mean, float value_norm_var, float* value, void* stream)
{
    int rc = check_gnet(critic);
    if (rc) return rc;
    rc = check_rows(critic, x, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!value || critic->out_dim != 1) return em::fail(EM_ERR_INVALID, "em_gmlp_value: needs out_dim 1 and an output");
    if (n_rows == 0) return EM_OK;
    GArgs ga;
    fill_common(ga, critic, x, ldx, n_rows, group_off, perm, split);
    ga.value = value; ga.norm_mean = value_norm_mean; ga.norm_scale = sqrtf(value_norm_var + 1e-8f);
    return launch_gfwd<MODE_VALUE>(ga, (hipStream_t)stream);
}

int64_t em_gmlp_slab_floats(int32_t in_dim, int32_t out_dim, int is_actor)
{
    return (int64_t)HID * in_dim + HID + HID * HID + HID + (int64_t)out_dim * HID + out_dim + (is_actor ? out_dim : 0) + 3;
}

int em_gmlp_actor_grad(const em_gmlp* actor, const float* log_std, const float* x, int64_t ldx, const float* action,
                       const float* old_logp, const float* adv, const double* moments3, int64_t n_rows,
                       const int64_t* group_off, const int32_t* perm, int32_t split, float clip, float entropy_coeff,
                       float* slab, float* grad_out, float* loss3_out, void* stream)
{
    int rc = check_gnet(actor);
    if (rc) return rc;
    rc = check_rows(actor, x, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!log_std || !action || !old_logp || !adv || !moments3 || !slab || !grad_out || !loss3_out)
        return em::fail(EM_ERR_INVALID, "em_gmlp_actor_grad: null argument");
    GArgs ga;
    fill_common(ga, actor, x, ldx, n_rows, group_off, perm, split);
    ga.log_std = log_std; ga.action = action; ga.old_logp = old_logp; ga.adv = adv; ga.moments = moments3;
    ga.clip = clip; ga.ecoef = entropy_coeff; ga.slab = slab;
    ga.slab_stride = em_gmlp_slab_floats(actor->in_dim, actor->out_dim, 1);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_gbwd<true>(ga, s);
    if (rc) return rc;
    reduce_group_slabs(slab, split * 4, ga.slab_stride, actor->n_groups, grad_out, loss3_out, s);
    return em::hip_check(hipGetLastError(), "gmlp actor grad launch");
}

int em_gmlp_critic_grad(const em_gmlp* critic, const float* x, int64_t ldx, const float* ret, const double* moments3,
                        int64_t n_rows, const int64_t* group_off, const int32_t* perm, int32_t split, float value_loss_coeff,
                        float value_norm_mean, float value_norm_var, float* slab, float* grad_out, float* loss3_out,
                        void* stream)
{
    int rc = check_gnet(critic);
    if (rc) return rc;
    rc = check_rows(critic, x, ldx, n_rows, group_off, split);
    if (rc) return rc;
    if (!ret || !moments3 || !slab || !grad_out || !loss3_out || critic->out_dim != 1)
        return em::fail(EM_ERR_INVALID, "em_gmlp_critic_grad: bad argument");
    GArgs ga;
    fill_common(ga, critic, x, ldx, n_rows, group_off, perm, split);
    ga.ret = ret; ga.moments = moments3; ga.vcoef = value_loss_coeff; ga.norm_mean = value_norm_mean;
    ga.norm_scale = sqrtf(value_norm_var + 1e-8f); ga.slab = slab;
    ga.slab_stride = em_gmlp_slab_floats(critic->in_dim, 1, 0);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_gbwd<false>(ga, s);
    if (rc) return rc;
    reduce_group_slabs(slab, split * 4, ga.slab_stride, critic->n_groups, grad_out, loss3_out, s);
    return em::hip_check(hipGetLastError(), "gmlp critic grad launch");
}

int em_group_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n_per_group, int64_t n_groups,
                  double lr, int64_t step, double beta1, double beta2, double eps, double max_grad_norm, float* norms_out,
                  void* stream)
{
    if (!param || !grad || !exp_avg || !exp_avg_sq || n_per_group < 1 || n_groups < 1 || step < 1)
        return em::fail(EM_ERR_INVALID, "em_group_adam: bad argument");
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    GroupAdam a = {param, grad, exp_avg, exp_avg_sq, n_per_group, (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2,
                   (float)(1.0 - beta2), (float)eps, (float)max_grad_norm, (float)sqrt(bc2), 0.0f};
    int64_t parts = (n_per_group + 8191) / 8192;
    parts = parts < 1 ? 1 : (parts > 128 ? 128 : parts);
    hipLaunchKernelGGL(group_adam_kernel, dim3((unsigned)(n_groups * parts)), dim3(256), 0, (hipStream_t)stream, a, (int)parts, norms_out);
    return em::hip_check(hipGetLastError(), "group adam launch");
}

int em_group_moments(const float* adv, const int64_t* group_off, const int32_t* perm, int32_t n_groups, double* moments3,
                     void* stream)
{
    if (!adv || !group_off || !moments3 || n_groups < 1) return em::fail(EM_ERR_INVALID, "em_group_moments: bad argument");
    hipLaunchKernelGGL(group_moments_kernel, dim3((unsigned)n_groups), dim3(256), 0, (hipStream_t)stream, adv, group_off, perm,
                       moments3);
    return em::hip_check(hipGetLastError(), "group moments launch");
}

int em_gae_agents(int64_t n_envs, int32_t horizon, int32_t n_agents, const double* reward_tb, const uint8_t* done_tb,
                  const uint8_t* k_tb, const int32_t* first_tb, const int32_t* ent_agent, const float* ent_value, double gamma,
                  double lam, double* chain_state, float* ent_adv, float* ent_ret, void* stream)
{
    if (!reward_tb || !done_tb || !k_tb || !first_tb || !ent_agent || !ent_value || !chain_state || !ent_adv || !ent_ret)
        return em::fail(EM_ERR_INVALID, "em_gae_agents: null argument");
    if (n_envs < 1 || horizon < 1 || n_agents < 1) return em::fail(EM_ERR_INVALID, "em_gae_agents: bad sizes");
    const dim3 grid((unsigned)((n_envs * GAE_SUB + 255) / 256));
    const size_t lds = (size_t)(256 / GAE_SUB) * 2 * (size_t)n_agents * sizeof(double);
    if (lds <= 64 * 1024)
        hipLaunchKernelGGL(gae_agents_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, n_envs, horizon, n_agents, reward_tb,
                           done_tb, k_tb, first_tb, ent_agent, ent_value, gamma, lam, chain_state, ent_adv, ent_ret);
    else
        hipLaunchKernelGGL(gae_agents_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, n_envs, horizon, n_agents, reward_tb,
                           done_tb, k_tb, first_tb, ent_agent, ent_value, gamma, lam, chain_state, ent_adv, ent_ret);
    return em::hip_check(hipGetLastError(), "gae agents launch");
}

}  // extern "C"
