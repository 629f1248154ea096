// central_critic.hip - MAPPO's central critic on the scaled topology (MAPPOCritic, mappo_algorithm.py:74-90, fed the
// concatenated state of mappo_trainer.py:101-123) for D decider agents: state width D*(S+4) = 17,408 at 64 x 256.
//
// The dense state is >99 % zeros (about 2 of 256 deciders hold a task), so its first layer is evaluated SPARSELY:
//     fc1(state) = b1 + sum over the step's entries e of  W1[:, d(e)*W : (d(e)+1)*W] . obs_e          (W = S+4)
// which is exactly the dense product because absent deciders contribute zero rows (mappo_trainer.py:113-122).
//   cc_l1_fwd_kernel   per entry, grouped by decider (one 64 x W weight block per workgroup, as gmlp.hip):  P[e] = W1_d . obs_e
//   cc_head_fwd_kernel per (step, env) state:  h1 = relu(b1 + sum_j P[first + j]);  h2 = relu(W2 h1 + b2);  v = W3 h2 + b3
//   cc_head_bwd_kernel forward recompute + value loss (mappo_algorithm.py:303,312-314; the state's k transitions share
//                      one evaluation, as em_ppo_critic_grad) + backward to dZ1 = dL/d(fc1 pre-activation) per state,
//                      gradients of b1, W2, b2, W3, b3 as slab rows
//   cc_l1_bwd_kernel   per decider:  dW1_d += sum over its entries of dZ1[state(e)] (x) obs_e
// All products run on v_mfma_f32_16x16x4_f32 in the transposed formulation of mappo_fwd.hip; summations have fixed
// order (slab rows + ordered reduction), no float atomics.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "edgemarl.h"
#include "em_common.h"
#include "em_lanes.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int HID = 64;
constexpr int TS = 17;
constexpr int S2 = 66;
constexpr int KU = 5;                 // observation width padded to 80
constexpr int K1P = KU * 16;
constexpr int S1 = (K1P + 2) | 2;

struct CCArgs {
    em_central_critic net;
    const float* ent_obs;      // [E][W]
    const int32_t* ent_state;  // [E] state row of the entry
    const int64_t* group_off;  // [D + 1] positions of each decider's entries in perm
    const int32_t* perm;       // position -> entry
    int32_t split, pad0;
    float* P;                  // [E][64] first-layer partial products
    int64_t n_states;
    const uint8_t* k_arr;      // [n_states] entries of the state
    const int32_t* first_arr;  // [n_states] first entry
    float* value_out;          // [n_states]
    float norm_mean, norm_scale;
    const float* ent_ret;      // [E]
    float* DZ;                 // [n_states][64]
    float* slab;
    int64_t slab_stride;
    float inv_L, vcoef;
};

__device__ inline void load_obs_frag(const float* __restrict__ obs, int W, int64_t e, bool valid, int q, float (&xb)[KU * 4])
{
    const float* xr = obs + (valid ? e : 0) * (int64_t)W;
    const bool vec_ok = (W & 3) == 0;
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = 16 * u + 4 * q;
        if (vec_ok && k + 3 < W) {
            const float4 v = *reinterpret_cast<const float4*>(xr + k);
            xb[u * 4 + 0] = v.x; xb[u * 4 + 1] = v.y; xb[u * 4 + 2] = v.z; xb[u * 4 + 3] = v.w;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) xb[u * 4 + r] = (k + r) < W ? xr[k + r] : 0.0f;
        }
    }
    if (!valid) {
#pragma unroll
        for (int s = 0; s < KU * 4; ++s) xb[s] = 0.0f;
    }
}

// P[e][j] = sum_i W1[j][d*W + i] * obs[e][i] for the entries of decider d = blockIdx.x / split
__global__ __launch_bounds__(256) void cc_l1_fwd_kernel(const CCArgs a)
{
    __shared__ float W1s[HID * S1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int W = a.net.width;
    const int g = blockIdx.x / a.split, part = blockIdx.x - g * a.split;
    const int64_t r0 = a.group_off[g], n_rows = a.group_off[g + 1] - r0, n_tiles = (n_rows + 15) >> 4;
    if ((int64_t)part * 4 >= n_tiles) return;
    const float* w1 = a.net.w1 + (int64_t)g * W;
    for (int i = threadIdx.x; i < HID * K1P; i += 256) {
        const int r = i / K1P, k = i - r * K1P;
        W1s[r * S1 + k] = k < W ? w1[(int64_t)r * a.net.ldw + k] : 0.0f;
    }
    __syncthreads();
    // Two dependent memory round trips feed a tile (permutation, then the scattered observation row) and one or two
    // wavefronts per SIMD hide neither: the entry numbers are requested two tiles ahead and the rows one tile ahead, so that
    // a tile's 80 MFMAs never wait for memory (10 ms -> 4 ms per launch over 36.7 M entries together with two workgroups
    // per decider).
    const int64_t stride = (int64_t)a.split * 4;
    auto entry_of = [&](int64_t tl) -> int32_t {
        const int64_t row = tl * 16 + c;
        return (tl < n_tiles && row < n_rows) ? a.perm[r0 + row] : -1;
    };
    int64_t tile = (int64_t)part * 4 + wave;
    int32_t e0 = entry_of(tile), e1 = entry_of(tile + stride);
    float xb[KU * 4];
    load_obs_frag(a.ent_obs, W, e0 < 0 ? 0 : e0, e0 >= 0, q, xb);
    // A tile's partial rows are stored at the START of the next tile, right behind that tile's look-ahead loads: stores count
    // in the same in-order queue as loads, and stored at the end of their own tile they were the youngest entries when the
    // next tile waited for its operands - every tile then waited for its predecessor's stores to be acknowledged.
    f32x4 hp[4];
    int32_t ep = -1;
    auto store_rows = [&](int32_t e, const f32x4 (&h)[4]) {
        if (e >= 0) {
            float4* dst = reinterpret_cast<float4*>(a.P + (int64_t)e * HID);
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[4 * t + q] = make_float4(h[t][0], h[t][1], h[t][2], h[t][3]);   // j = 16t + 4q + r
        }
    };
    for (; tile < n_tiles; tile += stride) {
        const int32_t e2 = entry_of(tile + 2 * stride);
        float xn[KU * 4];
        load_obs_frag(a.ent_obs, W, e1 < 0 ? 0 : e1, e1 >= 0, q, xn);
        store_rows(ep, hp);
        f32x4 h[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KU * 4; ++s) {
            const int k = 16 * (s >> 2) + 4 * q + (s & 3);
#pragma unroll
            for (int t = 0; t < 4; ++t) h[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1s[(16 * t + c) * S1 + k], xb[s], h[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) hp[t] = h[t];
        ep = e0;
        e0 = e1; e1 = e2;
#pragma unroll
        for (int s = 0; s < KU * 4; ++s) xb[s] = xn[s];
    }
    store_rows(ep, hp);
}

// LDS of the head kernels: W2 [64][S2] | W3 [16][S2] (row 0 used) | b1[64] b2[64] b3[16]
enum { H_W2 = 0, H_W3 = H_W2 + HID * S2, H_B1 = H_W3 + 16 * S2, H_B2 = H_B1 + HID, H_B3 = H_B2 + HID, H_END = H_B3 + 16 };

__device__ inline void stage_head(float* lds, const em_central_critic& n)
{
    for (int i = threadIdx.x; i < HID * HID; i += 256) lds[H_W2 + (i >> 6) * S2 + (i & 63)] = n.w2[i];
    for (int i = threadIdx.x; i < 16 * HID; i += 256) lds[H_W3 + (i >> 6) * S2 + (i & 63)] = (i >> 6) < 1 ? n.w3[i] : 0.0f;
    if (threadIdx.x < HID) { lds[H_B1 + threadIdx.x] = n.b1[threadIdx.x]; lds[H_B2 + threadIdx.x] = n.b2[threadIdx.x]; }
    if (threadIdx.x < 16) lds[H_B3 + threadIdx.x] = threadIdx.x == 0 ? n.b3[0] : 0.0f;
}

// h1[t][r] = relu(b1 + sum of the state's partial products)[feature 16t + 4q + r] for state `c` of the tile: the
// accumulator layout of mappo_fwd.hip, which is also the B operand of the next layer.
__device__ inline void gather_h1(const CCArgs& a, const float* lds, int k, int64_t first, int q, f32x4 (&h1)[4])
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = lds[H_B1 + 16 * t + 4 * q + r];
    int kmax = k;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) { const int o = __shfl_xor(kmax, m); kmax = o > kmax ? o : kmax; }
    // One entry ahead: the loads of entry j + 1 are in flight while entry j is added, so the state's k partial rows cost
    // one memory round trip in all instead of one each (the sum keeps its order: entry 0, 1, ...).
    float4 v[4], vn[4];
    auto fetch = [&](int j, float4 (&dst)[4]) {
        if (j < k) {
            const float4* src = reinterpret_cast<const float4*>(a.P + (first + j) * HID);
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[t] = src[4 * t + q];
        }
    };
    fetch(0, v);
    for (int j = 0; j < kmax; ++j) {
        fetch(j + 1, vn);
        if (j < k) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { h1[t][0] += v[t].x; h1[t][1] += v[t].y; h1[t][2] += v[t].z; h1[t][3] += v[t].w; }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = vn[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h1[t][r] = fmaxf(h1[t][r], 0.0f);
}

// Layer 2 on the matrix cores (the fragments of the next reduction index are requested before the MFMAs of this one: the
// compiler's own schedule waited for LDS after every pair of words), the one-output head on the vector ALU: v = b3 + w3 . h2
// from the sixteen h2 values the lane holds (four partial sums, two row exchanges), on every lane of the row, instead of a
// chain of sixteen dependent MFMAs of which 1/16 is used (as the critics of mappo_bwd.hip / gmlp.hip).
__device__ inline void head_forward(const float* lds, int q, int c, const f32x4 (&h1)[4], f32x4 (&h2)[4], float& y0)
{
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = lds[H_B2 + 16 * t + 4 * q + r];
    {
        const float* wrow = lds + H_W2 + c * S2 + 4 * q;
        float fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = wrow[16 * t * S2];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            float (&cur_)[4] = (s & 1) ? fb : fa;
            float (&nxt_)[4] = (s & 1) ? fa : fb;
            if (s + 1 < 16) {
                const int k1 = 16 * ((s + 1) >> 2) + ((s + 1) & 3);
#pragma unroll
                for (int t = 0; t < 4; ++t) nxt_[t] = wrow[16 * t * S2 + k1];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur_[t], h1[s >> 2][s & 3], h2[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float ps[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) h2[t][r] = fmaxf(h2[t][r], 0.0f);
        const f32x4 w = *reinterpret_cast<const f32x4*>(lds + H_W3 + 16 * t + 4 * q);
        ps[t] = w[0] * h2[t][0];
#pragma unroll
        for (int r = 1; r < 4; ++r) ps[t] = fmaf(w[r], h2[t][r], ps[t]);
    }
    y0 = em_rows_sum((ps[0] + ps[1]) + (ps[2] + ps[3])) + lds[H_B3];
}

__global__ __launch_bounds__(256) void cc_head_fwd_kernel(const CCArgs a)
{
    __shared__ float lds[H_END];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    stage_head(lds, a.net);
    __syncthreads();
    const int64_t n_tiles = (a.n_states + 15) >> 4;
    // a state's {entries, first entry} one tile ahead: the gather's row addresses are then known at the top of the tile
    int k_n = 0, first_n = 0;
    {
        const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
        if (s0 < a.n_states) { k_n = (int)a.k_arr[s0]; first_n = a.first_arr[s0]; }
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t s = tile * 16 + c;
        const bool valid = s < a.n_states;
        const int k = k_n;
        const int64_t first = first_n;
        k_n = 0;
        {
            const int64_t sn = s + (int64_t)gridDim.x * 64;
            if (sn < a.n_states) { k_n = (int)a.k_arr[sn]; first_n = a.first_arr[sn]; }
        }
        f32x4 h1[4], h2[4];
        float y0;
        gather_h1(a, lds, k, first, q, h1);
        head_forward(lds, q, c, h1, h2, y0);
        if (valid && k > 0 && q == 0) a.value_out[s] = y0 * a.norm_scale + a.norm_mean;   // denormalize, mappo_algorithm.py:384-386
    }
}

// slab row of the head: [b1(64) | w2(4096) | b2(64) | w3(64) | b3(1) | loss, 0, count]
constexpr int HS_B1 = 0, HS_W2 = HID, HS_B2 = HS_W2 + HID * HID, HS_W3 = HS_B2 + HID, HS_B3 = HS_W3 + HID, HS_LOSS = HS_B3 + 1,
              HS_STRIDE = HS_LOSS + 3;

__global__ __launch_bounds__(256, 2) void cc_head_bwd_kernel(const CCArgs a)
{
    __shared__ float lds[H_END + 4 * HID * TS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    float* scr = lds + H_END + wave * (HID * TS);
    stage_head(lds, a.net);
    __syncthreads();
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gW2[4][4], gW3[4], gb1[4], gb2[4];
    float gb3 = 0.0f, cnt_sum = 0.0f;
    double loss_sum = 0.0;          // float64 per lane (see mappo_bwd.hip)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) gW2[t][u] = zero4;
        gW3[t] = zero4; gb1[t] = zero4; gb2[t] = zero4;
    }
    const int64_t n_tiles = (a.n_states + 15) >> 4;
    // a state's {entries, first entry} one tile ahead: the gather's row addresses are then known at the top of the tile
    int k_n = 0, first_n = 0;
    {
        const int64_t s0 = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
        if (s0 < a.n_states) { k_n = (int)a.k_arr[s0]; first_n = a.first_arr[s0]; }
    }
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t s = tile * 16 + c;
        const bool valid = s < a.n_states;
        const int k = k_n;
        const int64_t first = first_n;
        k_n = 0;
        {
            const int64_t sn = s + (int64_t)gridDim.x * 64;
            if (sn < a.n_states) { k_n = (int)a.k_arr[sn]; first_n = a.first_arr[sn]; }
        }
        // the returns of the state's first KPRE transitions are requested here, unconditionally, and used after the forward
        // pass (read inside the loss, each was a dependent memory round trip on the lanes that evaluate it)
        constexpr int KPRE = 4;
        float p_ret[KPRE];
#pragma unroll
        for (int j = 0; j < KPRE; ++j) p_ret[j] = a.ent_ret[first + (j < k ? j : 0)];
        f32x4 h1[4], h2[4];
        float y0;
        gather_h1(a, lds, k, first, q, h1);
        head_forward(lds, q, c, h1, h2, y0);
        float dy0 = 0.0f;
        if (valid && k > 0 && q == 0) {
            float gsum = 0.0f;
            const float vn = (y0 - a.norm_mean) / a.norm_scale;
#pragma unroll
            for (int j = 0; j < KPRE; ++j) {       // the state's k transitions share its value (mappo_algorithm.py:303,312-314)
                if (j < k) {
                    const float e = vn - (p_ret[j] - a.norm_mean) / a.norm_scale;
                    gsum += e;
                    loss_sum += (double)(e * e);
                }
            }
            for (int j = KPRE; j < k; ++j) {
                const float e = vn - (a.ent_ret[first + j] - a.norm_mean) / a.norm_scale;
                gsum += e;
                loss_sum += (double)(e * e);
            }
            cnt_sum += (float)k;
            dy0 = a.vcoef * a.inv_L * gsum / a.norm_scale;
            gb3 += dy0;
        }
        // value head on the vector ALU: dH2 = w3 * dv through the ReLU; dW3 accumulates dv * h2 per lane (row c's share, the
        // sixteen rows' lanes are summed when the slab row is written)
        const float dv = em_rows_sum(dy0);             // only the lane with q = 0 holds dL/dv of row c, the others 0
        f32x4 dz2[4], dz1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(lds + H_W3 + 16 * t + 4 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gW3[t][r] = fmaf(dv, h2[t][r], gW3[t][r]);
                dz2[t][r] = h2[t][r] > 0.0f ? w[r] * dv : 0.0f;
            }
            dz1[t] = zero4;
            gb2[t] += dz2[t];
        }
        {   // dH1^T = W2^T . dZ2^T; fragments one reduction index ahead
            const float* wcol = lds + H_W2 + (4 * q) * S2 + c;
            float fa[4], fb[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) fa[t] = wcol[16 * t];
#pragma unroll
            for (int sidx = 0; sidx < 16; ++sidx) {
                float (&cur_)[4] = (sidx & 1) ? fb : fa;
                float (&nxt_)[4] = (sidx & 1) ? fa : fb;
                if (sidx + 1 < 16) {
                    const int k1 = 16 * ((sidx + 1) >> 2) + ((sidx + 1) & 3);
#pragma unroll
                    for (int t = 0; t < 4; ++t) nxt_[t] = wcol[k1 * S2 + 16 * t];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t) dz1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur_[t], dz2[sidx >> 2][sidx & 3], dz1[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[t][r] = h1[t][r] > 0.0f ? dz1[t][r] : 0.0f;
            gb1[t] += dz1[t];
        }
        if (valid && k > 0) {                      // dL/d(fc1 pre-activation) of the state, consumed per entry by cc_l1_bwd_kernel
            float4* dst = reinterpret_cast<float4*>(a.DZ + s * HID);
#pragma unroll
            for (int t = 0; t < 4; ++t) dst[4 * t + q] = make_float4(dz1[t][0], dz1[t][1], dz1[t][2], dz1[t][3]);
        }
        // weight gradients of layers 3 and 2 (reduction over the tile's 16 states through LDS transposes)
        float tr_a[4][4], tr_b[4][4];
        auto transpose64 = [&](const f32x4(&m)[4], float(&out)[4][4]) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) scr[(16 * t + 4 * q + r) * TS + c] = m[t][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) out[t][sx] = scr[(16 * t + c) * TS + 4 * sx + q];
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };
        transpose64(dz2, tr_a);
        transpose64(h1, tr_b);
#pragma unroll
        for (int sx = 0; sx < 4; ++sx)
#pragma unroll
            for (int t2 = 0; t2 < 4; ++t2)
#pragma unroll
                for (int t1 = 0; t1 < 4; ++t1)
                    gW2[t2][t1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_a[t2][sx], tr_b[t1][sx], gW2[t2][t1], 0, 0, 0);
    }
    float* out = a.slab + ((int64_t)blockIdx.x * 4 + wave) * a.slab_stride;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = 16 * t + 4 * q + r;
#pragma unroll
            for (int t1 = 0; t1 < 4; ++t1) out[HS_W2 + j * HID + 16 * t1 + c] = gW2[t][t1][r];
        }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                  // gW3[t][r]: this lane's (row c) share of dW3[0][16t + 4q + r]
            float v = gW3[t][r];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
            if (c == 0) out[HS_W3 + 16 * t + 4 * q + r] = v;
        }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gb1[t][r] += __shfl_xor(gb1[t][r], m);
                gb2[t][r] += __shfl_xor(gb2[t][r], m);
            }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        loss_sum += __shfl_xor(loss_sum, m);
        cnt_sum += __shfl_xor(cnt_sum, m);
        gb3 += __shfl_xor(gb3, m);
    }
    if (c == 0) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                out[HS_B1 + 16 * t + 4 * q + r] = gb1[t][r];
                out[HS_B2 + 16 * t + 4 * q + r] = gb2[t][r];
            }
    }
    if (lane == 0) { out[HS_B3] = gb3; out[HS_LOSS] = (float)loss_sum; out[HS_LOSS + 1] = 0.0f; out[HS_LOSS + 2] = cnt_sum; }
}

// dW1[j][d*W + i] += sum over the entries e of decider d of dZ1[state(e)][j] * obs[e][i]; one slab row [64][W] per wavefront
__global__ __launch_bounds__(256) void cc_l1_bwd_kernel(const CCArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, c = lane & 15;
    const int W = a.net.width;
    const int g = blockIdx.x / a.split, part = blockIdx.x - g * a.split;
    const int64_t r0 = a.group_off[g], n_rows = a.group_off[g + 1] - r0, n_tiles = (n_rows + 15) >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 gW1[4][KU];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < KU; ++u) gW1[t][u] = zero4;
    // Three dependent round trips feed a tile (permutation -> state of the entry -> dZ1 row; the observation row hangs off
    // the first): entry numbers three tiles ahead, their states two, the operand values one, so the 80 MFMAs of a tile
    // find their operands in registers.
    const int64_t stride = (int64_t)a.split * 4;
    auto entries_of = [&](int64_t tl, int32_t (&e)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rb = tl * 16 + 4 * s + q;            // batch index on the reduction axis
            e[s] = (tl < n_tiles && rb < n_rows) ? a.perm[r0 + rb] : -1;
        }
    };
    auto states_of = [&](const int32_t (&e)[4], int32_t (&st)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) st[s] = e[s] >= 0 ? a.ent_state[e[s]] : 0;
    };
    auto values_of = [&](const int32_t (&e)[4], const int32_t (&st)[4], float (&av)[4][4], float (&xv)[4][KU]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool vb = e[s] >= 0;
            const float* dz = a.DZ + (int64_t)st[s] * HID;
            const float* xrow = a.ent_obs + (int64_t)(vb ? e[s] : 0) * W;
#pragma unroll
            for (int t = 0; t < 4; ++t) av[s][t] = vb ? dz[16 * t + c] : 0.0f;       // dZ1^T[j = 16t + c][batch]
#pragma unroll
            for (int u = 0; u < KU; ++u) { const int i = 16 * u + c; xv[s][u] = (vb && i < W) ? xrow[i] : 0.0f; }
        }
    };
    int64_t tile = (int64_t)part * 4 + wave;
    int32_t e1[4], e2[4], e3[4], st1[4], st2[4];
    float av[4][4], xv[4][KU];
    {
        int32_t e0[4], st0[4];
        entries_of(tile, e0); entries_of(tile + stride, e1); entries_of(tile + 2 * stride, e2);
        states_of(e0, st0); states_of(e1, st1);
        values_of(e0, st0, av, xv);
    }
    for (; tile < n_tiles; tile += stride) {
        entries_of(tile + 3 * stride, e3);
        states_of(e2, st2);
        float an[4][4], xn[4][KU];
        values_of(e1, st1, an, xn);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) gW1[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][t], xv[s][u], gW1[t][u], 0, 0, 0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            e1[s] = e2[s]; e2[s] = e3[s]; st1[s] = st2[s];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[s][t] = an[s][t];
#pragma unroll
            for (int u = 0; u < KU; ++u) xv[s][u] = xn[s][u];
        }
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
                if (i < W) out[j * W + i] = gW1[t][u][r];
            }
        }
}

// grad_w1[j][d*W + i] = sum of decider d's slab rows (fixed order)
__global__ __launch_bounds__(256) void cc_reduce_w1_kernel(const float* __restrict__ slab, int rows_per_group, int W, int ldw,
                                                           float* __restrict__ grad_w1)
{
    const int g = blockIdx.y;
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HID * W) return;
    const float* s = slab + (int64_t)g * rows_per_group * HID * W + p;
    float acc = 0.0f;
    for (int r = 0; r < rows_per_group; ++r) acc += s[(int64_t)r * HID * W];
    const int j = p / W, i = p - j * W;
    grad_w1[(int64_t)j * ldw + (int64_t)g * W + i] = acc;
}

// head gradients: [b1 | w2 | b2 | w3 | b3] and the 3 loss sums from the head slab rows (fixed order)
__global__ __launch_bounds__(256) void cc_reduce_head_kernel(const float* __restrict__ slab, int n_rows, float* __restrict__ grad_tail,
                                                             float* __restrict__ loss3)
{
    // 32 columns x 8 row lanes per workgroup: row lane j sums the rows r = j (mod 8) in ascending order, the eight partial
    // sums are combined in the order j = 0..7 - a fixed order, and 135 workgroups instead of the 17 that summed 2,048 rows
    // one after the other (0.64 ms per epoch)
    __shared__ float part[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int p = blockIdx.x * 32 + cl;
    float acc = 0.0f;
    if (p < HS_STRIDE) {
        const float* s = slab + p;
        int r = rl;
        for (; r + 24 < n_rows; r += 32) {
            const float v0 = s[(int64_t)r * HS_STRIDE], v1 = s[(int64_t)(r + 8) * HS_STRIDE], v2 = s[(int64_t)(r + 16) * HS_STRIDE],
                        v3 = s[(int64_t)(r + 24) * HS_STRIDE];
            acc = (((acc + v0) + v1) + v2) + v3;
        }
        for (; r < n_rows; r += 8) acc += s[(int64_t)r * HS_STRIDE];
    }
    part[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && p < HS_STRIDE) {
        float t = part[0][cl];
#pragma unroll
        for (int j = 1; j < 8; ++j) t += part[j][cl];
        if (p < HS_LOSS) grad_tail[p] = t;
        else loss3[p - HS_LOSS] = t;
    }
}

int check_cc(const em_central_critic* n)
{
    if (!n || !n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->w3 || !n->b3) return em::fail(EM_ERR_INVALID, "central critic: null parameter");
    if (n->width < 1 || n->width > K1P || n->n_deciders < 1 || n->n_deciders > EMS_MAX_DECIDERS || n->ldw < n->n_deciders * n->width)
        return em::fail(EM_ERR_INVALID, "central critic: unsupported shape (width <= 80)");
    return EM_OK;
}

int head_blocks(int64_t n_states)
{
    const int64_t tiles = (n_states + 15) / 16;
    int64_t b = (tiles + 3) / 4;
    if (b > 512) b = 512;
    return (int)(b < 1 ? 1 : b);
}

}  // namespace

extern "C" {

int em_cc_forward(const em_central_critic* net, const float* ent_obs, const int64_t* group_off, const int32_t* perm, int32_t split,
                  float* partial, int64_t n_states, const uint8_t* k_arr, const int32_t* first_arr, float value_norm_mean,
                  float value_norm_var, float* value_out, void* stream)
{
    int rc = check_cc(net);
    if (rc) return rc;
    if (!ent_obs || !group_off || !perm || !partial || !k_arr || !first_arr || !value_out || split < 1 || n_states < 1)
        return em::fail(EM_ERR_INVALID, "em_cc_forward: bad argument");
    CCArgs a;
    std::memset(&a, 0, sizeof a);
    a.net = *net; a.ent_obs = ent_obs; a.group_off = group_off; a.perm = perm; a.split = split; a.P = partial;
    a.n_states = n_states; a.k_arr = k_arr; a.first_arr = first_arr; a.value_out = value_out;
    a.norm_mean = value_norm_mean; a.norm_scale = sqrtf(value_norm_var + 1e-8f);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(cc_l1_fwd_kernel, dim3((unsigned)(net->n_deciders * split)), dim3(256), 0, s, a);
    // 92 registers: four workgroups per CU hide the gather of the partial rows (the backward kernel's slab fixes ITS grid)
    const int64_t fwd_tiles = (n_states + 15) / 16;
    int64_t fwd_blocks = (fwd_tiles + 3) / 4;
    if (fwd_blocks > 1024) fwd_blocks = 1024;
    hipLaunchKernelGGL(cc_head_fwd_kernel, dim3((unsigned)(fwd_blocks < 1 ? 1 : fwd_blocks)), dim3(256), 0, s, a);
    return em::hip_check(hipGetLastError(), "central critic forward launch");
}

int64_t em_cc_scratch_floats(const em_central_critic* net, int32_t split, int64_t n_states)
{
    if (!net) return -1;
    const int64_t l1 = (int64_t)net->n_deciders * split * 4 * HID * net->width;
    const int64_t head = (int64_t)head_blocks(n_states) * 4 * HS_STRIDE;
    return l1 > head ? l1 : head;
}

int em_cc_grad(const em_central_critic* net, const float* ent_obs, const int32_t* ent_state, const float* ent_ret,
               const int64_t* group_off, const int32_t* perm, int32_t split, float* partial, int64_t n_states,
               const uint8_t* k_arr, const int32_t* first_arr, int64_t n_global, float value_loss_coeff, float value_norm_mean,
               float value_norm_var, float* dz, float* slab, float* grad_out, float* loss3_out, void* stream)
{
    int rc = check_cc(net);
    if (rc) return rc;
    if (!ent_obs || !ent_state || !ent_ret || !group_off || !perm || !partial || !k_arr || !first_arr || !dz || !slab || !grad_out ||
        !loss3_out || split < 1 || n_states < 1 || n_global < 1)
        return em::fail(EM_ERR_INVALID, "em_cc_grad: bad argument");
    CCArgs a;
    std::memset(&a, 0, sizeof a);
    a.net = *net; a.ent_obs = ent_obs; a.ent_state = ent_state; a.ent_ret = ent_ret; a.group_off = group_off; a.perm = perm;
    a.split = split; a.P = partial; a.n_states = n_states; a.k_arr = k_arr; a.first_arr = first_arr; a.DZ = dz; a.slab = slab;
    a.norm_mean = value_norm_mean; a.norm_scale = sqrtf(value_norm_var + 1e-8f);
    a.inv_L = 1.0f / (float)n_global; a.vcoef = value_loss_coeff;
    hipStream_t s = (hipStream_t)stream;
    const int D = net->n_deciders, W = net->width;
    const int hb = head_blocks(n_states);
    // forward partial products with the current weights, then the head (loss, dZ1, head gradients), then dW1 per decider
    hipLaunchKernelGGL(cc_l1_fwd_kernel, dim3((unsigned)(D * split)), dim3(256), 0, s, a);
    a.slab_stride = HS_STRIDE;
    hipLaunchKernelGGL(cc_head_bwd_kernel, dim3((unsigned)hb), dim3(256), 0, s, a);
    float* tail = grad_out + (int64_t)HID * net->ldw;          // [b1 | w2 | b2 | w3 | b3] follow w1
    hipLaunchKernelGGL(cc_reduce_head_kernel, dim3((HS_STRIDE + 31) / 32), dim3(256), 0, s, slab, hb * 4, tail, loss3_out);
    a.slab_stride = (int64_t)HID * W;
    hipLaunchKernelGGL(cc_l1_bwd_kernel, dim3((unsigned)(D * split)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(cc_reduce_w1_kernel, dim3((unsigned)((HID * W + 255) / 256), (unsigned)D), dim3(256), 0, s, slab, split * 4, W,
                       net->ldw, grad_out);
    return em::hip_check(hipGetLastError(), "central critic grad launch");
}

}  // extern "C"
