// mappo_bwd.hip - update-side MAPPO kernels for MI355X (gfx950):
//   * gae_kernel: reverse scan of MAPPO.compute_gae (mappo_algorithm.py:241-262) over each environment's
//     segment of the flattened per-agent transition list, float64, one lane per environment, plus the
//     moments of the advantages for the global normalisation (mappo_algorithm.py:293)
//   * mlp3_bwd_kernel: fused forward-recompute + PPO loss + backward of the actor (PPO-clip + entropy,
//     mappo_algorithm.py:302-309,317-320) or the critic (value MSE through the normaliser, :303,312-314)
//     on the f32 matrix cores; weight gradients accumulate in registers per wavefront and leave as one
//     slab row per wavefront (deterministic, no atomics)
//   * reduce_slabs_kernel, adam_kernel: slab reduction, clip_grad_norm_ (:328-329) and Adam (:331-332)
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "edgemarl.h"
#include "em_common.h"
#include "em_lanes.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int HID = 64;
constexpr int TS = 17;      // row stride of the transpose scratch (floats)
constexpr int S2 = 66;      // LDS row stride of the 64-wide weight matrices (2*odd: conflict-free A-fragment reads)

// ------------------------------------------------------------------------------------------------
// GAE
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gae_kernel(int64_t B, int T, const double* __restrict__ reward_tb,
                                                  const float* __restrict__ value_tb, const uint8_t* __restrict__ done_tb,
                                                  const uint8_t* __restrict__ k_tb, const int32_t* __restrict__ first_tb,
                                                  double gamma, double lam, float* __restrict__ adv,
                                                  float* __restrict__ ret, double* __restrict__ partials)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    if (b < B) {
        double gae = 0.0, v_next = 0.0;   // next_value = 0 closes every environment's segment (:284)
        // the step's five scalars are requested two steps ahead, unconditionally (a load under `if (k)` is waited for where
        // the branch ends, and one lane per environment has nothing else to do while it waits: 1.19 ms of dependent round
        // trips at 65,536 environments x 1000 steps)
        struct In { int k; double r; float v; uint8_t dn; int first; };
        auto fetch = [&](int t) {
            const int64_t i = (int64_t)(t < 0 ? 0 : t) * B + b;
            In o;
            o.k = k_tb[i]; o.r = reward_tb[i]; o.v = value_tb[i]; o.dn = done_tb[i]; o.first = first_tb[i];
            if (t < 0) o.k = 0;
            return o;
        };
        In cur = fetch(T - 1), nxt = fetch(T - 2);
        for (int t = T - 1; t >= 0; --t) {
            const In nn = fetch(t - 2);
            const int k = cur.k;
            const double r = cur.r, v = (double)cur.v;
            const double nd = 1.0 - (double)cur.dn;
            const int64_t first = cur.first;
            cur = nxt; nxt = nn;
            if (k == 0) continue;
            for (int j = k - 1; j >= 0; --j) {
                const double nv = (j == k - 1) ? v_next : v;      // same-step entries chain through each other
                const double delta = r + gamma * nv * nd - v;
                gae = delta + gamma * lam * nd * gae;
                const float a32 = (float)gae;
                adv[first + j] = a32;
                ret[first + j] = (float)(gae + v);
                s0 += 1.0; s1 += (double)a32; s2 += (double)a32 * (double)a32;
            }
            v_next = v;
        }
    }
    // block reduction of the moments (fixed order: deterministic)
    __shared__ double red[3][256];
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[0][threadIdx.x] += red[0][threadIdx.x + off];
            red[1][threadIdx.x] += red[1][threadIdx.x + off];
            red[2][threadIdx.x] += red[2][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partials[3 * blockIdx.x + 0] = red[0][0];
        partials[3 * blockIdx.x + 1] = red[1][0];
        partials[3 * blockIdx.x + 2] = red[2][0];
    }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partials, int64_t n_blocks,
                                                           double* __restrict__ moments)
{
    __shared__ double red[3][256];
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = threadIdx.x; i < n_blocks; i += 256)
        for (int c = 0; c < 3; ++c) s[c] += partials[3 * i + c];
    for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x < 3) moments[threadIdx.x] = red[threadIdx.x][0];
}

// Scratch of em_gae_static (the entry point that allocates nothing): the per-workgroup advantage moments that em_gae
// takes as its caller-owned `partials` argument, 3 doubles per 256 environments, EM_GAE_STATIC_MAX_ENVS environments at
// most = 96 KB that every GPU the library runs on holds until the process ends.  In host code this NAME is only a
// stand-in for the symbol; the device address comes from hipGetSymbolAddress (once per device, kept in
// g_gae_scratch_ptr) and only that pointer is ever handed to a kernel.
static __device__ __align__(256) double scratch[3 * (EM_GAE_STATIC_MAX_ENVS / 256)];
constexpr int GAE_MAX_DEVICES = 64;
static std::atomic<double*> g_gae_scratch_ptr[GAE_MAX_DEVICES];

// ------------------------------------------------------------------------------------------------
// fused forward-recompute + loss + backward
// ------------------------------------------------------------------------------------------------
struct BwdArgs {
    em_mlp3 net;
    const float* x;            // [rows][ldx]
    const int32_t* x_rows;     // optional: input row r is x[x_rows[r]] (states held by reference in the observation history)
    int64_t ldx, n_rows;
    float* slab;               // [n_waves][slab_stride]
    int64_t slab_stride;
    float inv_L;               // 1 / global number of transitions
    // actor
    const float* log_std;      // [O]
    const float* action;       // [rows][O]
    const float* old_logp;     // [rows]
    const float* adv;          // [rows] un-normalised
    const double* moments;     // {count, sum, sum of squares} of the advantages (global)
    float clip, ecoef;
    // critic
    const int32_t* st_first;   // [rows]
    const uint8_t* st_k;       // [rows]
    const float* ent_ret;      // [E]
    float vcoef, norm_mean, norm_scale;
};

// STAGE (wide inputs with contiguous, 16-byte aligned rows of K1 % 4 == 0 floats: the 140-wide critic): a tile's 16 input rows
// are one contiguous span; it is copied ONCE, global -> LDS with no register in between (global_load_lds_dwordx4), into a
// per-wavefront double buffer while the previous tile computes.  The forward reads its x fragments from that copy and the
// dW1 products read the same copy again, where the register version fetched 4 x KU dependent scalars per tile from
// global memory a second time; with one wavefront per SIMD nothing else hides either latency.
// Narrow inputs (KU == 1: the 14-wide actor) are compiled for TWO workgroups per CU (<= 256 registers): the loss epilogue,
// the LDS transposes and the accumulator drains of one wavefront then run under the other wavefront's MFMAs.  To fit, the
// layer biases are re-read from LDS per tile instead of living in 32 registers, and the bias gradients are row sums of the
// transposed dZ tiles that the weight-gradient products stage in LDS anyway (one register per lane instead of 32).
// SPLIT (STAGE kernels, 512 threads): wavefronts w and w + 4 share a SIMD and a tile sequence.  Wavefront w ("activation")
// runs the forward pass, the loss, dZ2/dZ1, dW3 and dW2; wavefront w + 4 ("gradient") holds the dW1 accumulators (144 of
// the critic's 224 accumulator registers, which is what kept the kernel at one wavefront per SIMD), issues the LDS-DMA
// copies of the pair's input tiles one tile ahead (the by-reference gather's index arithmetic was 2.7 k cycles at the top
// of every tile) and accumulates dW1 += dZ1^T X for the tile the activation wavefront finished last, from the transposed
// dZ1 in the pair's scratch and the staged input tile.  Its 16 x KU MFMAs per tile have no dependence on the activation
// wavefront's current tile, so they fill the matrix pipe while that one waits for LDS, evaluates the loss or drains
// accumulators.  Three counters per pair in LDS order the hand-overs (tiles staged, dZ1 published, dZ1 read); per
// accumulator the sums are taken in the same order as in the one-wavefront kernel, so the gradients are bitwise the same.
#ifdef EM_STAMP_BWD   // cycles per phase of the activation wavefront (workgroup 1, wavefront 0), summed over its tiles
__device__ unsigned long long bwd_stamps[2][16];
#define BSTAMP(i)                                                                                        \
    do {                                                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter();                                    \
        if ((i) > 0) bst_acc[i] += (unsigned)(now_ - bst_prev);                                          \
        else bst_acc[0] += 1;                                                                            \
        bst_prev = now_;                                                                                 \
    } while (0)
#else
// the phase boundaries of a tile double as scheduling fences in the critic: left free, the scheduler mixes the phases
// of the value-head variant and the kernel loses what the shorter chain gains (19.1 against 17.7 ms per epoch, measured)
#define BSTAMP(i) do { if constexpr (!ACTOR) __builtin_amdgcn_sched_barrier(0); } while (0)
#endif
template <int KU, bool ACTOR, bool STAGE = false, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 512 : 256, (KU <= 1 ? 2 : 1)) void mlp3_bwd_kernel(const BwdArgs a)
{
    static_assert(!SPLIT || STAGE, "the gradient wavefront reads the staged input tile");
    extern __shared__ __align__(16) float lds[];
    constexpr int NT = SPLIT ? 512 : 256;
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;   // wave: the pair's index (LDS regions, tiles, slab row)
    const bool grad_wave = SPLIT && (threadIdx.x >> 8) != 0;
    const int q = lane >> 4, c = lane & 15;
    const int K1 = a.net.in_dim, O = a.net.out_dim;
    constexpr int K1P = KU * 16;
    constexpr int S1 = (K1P + 2) | 2;                 // 2*odd >= K1P + 2 when K1P is a multiple of 16
    float* W1s = lds;                                 // [64][S1]
    float* W2s = W1s + HID * S1;                      // [64][S2]
    float* W3s = W2s + HID * S2;                      // [16][S2] (rows >= O are zero)
    float* Bs = W3s + 16 * S2;                        // [2][64] biases of layers 1 and 2
    float* scr = Bs + 2 * HID + wave * (HID * TS);    // per-wavefront transpose scratch [64][TS]
    constexpr int NI = (4 * K1P + 63) / 64;           // wavefront-wide 16-byte copies per tile (16 rows x K1 floats)
    constexpr int XBUF = NI * 256;                    // floats per staged tile, whole copies (the tail of the last is slack)
    float* xs = Bs + 2 * HID + 4 * (HID * TS) + wave * (2 * XBUF);
    // narrow inputs (register fragments, no staged copy): the tile's inputs transposed, [feature][row], for the dW1 product.
    // (Reading X[b][i] back from global memory in that product put an L2/HBM round trip in front of its MFMAs.)
    float* xtr = Bs + 2 * HID + 4 * (HID * TS) + wave * (K1P * TS);

    // SPLIT: {input tiles staged, -, dZ1 published, dZ1 read} counters of the pair, in tiles
    int* const flg = reinterpret_cast<int*>(Bs + 2 * HID + 4 * (HID * TS) + 4 * (2 * XBUF)) + 4 * wave;
    enum { F_XREADY = 0, F_READY = 2, F_SDONE = 3 };
    if (SPLIT && threadIdx.x < 16) reinterpret_cast<int*>(Bs + 2 * HID + 4 * (HID * TS) + 4 * (2 * XBUF))[threadIdx.x] = 0;

    for (int i = threadIdx.x; i < HID * K1P; i += NT) {
        const int r = i / K1P, k = i - r * K1P;
        W1s[r * S1 + k] = k < K1 ? a.net.w1[r * K1 + k] : 0.0f;
    }
    for (int i = threadIdx.x; i < HID * HID; i += NT) W2s[(i >> 6) * S2 + (i & 63)] = a.net.w2[i];
    for (int i = threadIdx.x; i < 16 * HID; i += NT) W3s[(i >> 6) * S2 + (i & 63)] = (i >> 6) < O ? a.net.w3[i] : 0.0f;
    if (threadIdx.x < 2 * HID) Bs[threadIdx.x] = threadIdx.x < HID ? a.net.b1[threadIdx.x] : a.net.b2[threadIdx.x - HID];
    __syncthreads();

    // hand-over waits.  Every wavefront of the workgroup is resident and both wavefronts of a pair walk the same tile
    // sequence, so each wait is answered; the spin is bounded all the same (a wait that runs out poisons the loss with NaN
    // and every later wait of the wavefront falls through), so that the grid always drains.
    bool poisoned = false;
    unsigned spin_acc[4] = {0, 0, 0, 0}; (void)spin_acc;
    auto wait_flag = [&](int idx, int want) {
        if (poisoned) return;
        int spins = 0;
        while (__hip_atomic_load(flg + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { poisoned = true; break; }
        }
#ifdef EM_STAMP_BWD
        spin_acc[idx] += (unsigned)spins;
#endif
    };
    auto post_flag = [&](int idx, int value) {
        if (lane == 0) __hip_atomic_store(flg + idx, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };

    f32x4 bias3;
#pragma unroll
    for (int r = 0; r < 4; ++r) bias3[r] = (4 * q + r) < O ? a.net.b3[4 * q + r] : 0.0f;
    const float b3_0 = a.net.b3[0];

    // gradient accumulators (per wavefront, whole launch)
    f32x4 gW1[4][KU], gW2[4][4], gW3[4], gb3, gls;
    float gb1_l = 0.0f, gb2_l = 0.0f;                 // bias gradients of feature `lane` (layers 1 and 2)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < KU; ++u) gW1[t][u] = zero4;
#pragma unroll
        for (int u = 0; u < 4; ++u) gW2[t][u] = zero4;
        gW3[t] = zero4;
    }
    gb3 = zero4; gls = zero4;
    // loss sums in float64 per lane: a float32 accumulator that adds one row's (nearly constant) entropy a few thousand times
    // rounds the same way every time and drifted 2e-5 relative at 54 M transitions (tests/test_fullsize_gpu.py)
    double loss_sum = 0.0;
    float cnt_sum = 0.0f;

    float adv_mean = 0.0f, adv_den = 1.0f, sd[4] = {1.f, 1.f, 1.f, 1.f}, ls[4] = {0.f, 0.f, 0.f, 0.f};
    float ivar[4] = {1.f, 1.f, 1.f, 1.f};     // 1 / sd^2: the log-density's quotients as products (<= 1 ulp per term)
    if constexpr (ACTOR) {
        const double n = a.moments[0], s1 = a.moments[1], s2 = a.moments[2];
        const double mean = s1 / n;
        double var = (s2 - s1 * s1 / n) / (n - 1.0);              // torch.std: unbiased
        if (!(var > 0.0)) var = 0.0;
        adv_mean = (float)mean;
        adv_den = (float)sqrt(var) + 1e-8f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < O) { ls[r] = a.log_std[4 * q + r]; sd[r] = expf(ls[r]); }
#pragma unroll
        for (int r = 0; r < 4; ++r) ivar[r] = 1.0f / (sd[r] * sd[r]);
    }

    float ent_row = 0.0f;                              // entropy of the state-independent Gaussian, the same for every row
    if constexpr (ACTOR) {
        for (int o = 0; o < O; ++o) ent_row += 0.5f + 0.91893853320467274f + a.log_std[o];
    }
    const bool vec_ok = ((a.ldx & 3) == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0);
    const int64_t n_tiles = (a.n_rows + 15) >> 4;
    // x fragment of a tile: row c of the tile, features 16u + 4q + r
    auto load_x = [&](int64_t tile_, float(&xb_)[KU * 4]) {
        const int64_t row_ = tile_ * 16 + c;
        const bool valid_ = row_ < a.n_rows;
        int64_t src_ = valid_ ? row_ : 0;
        if (a.x_rows) src_ = valid_ ? (int64_t)a.x_rows[src_] : 0;
        const float* xr = a.x + src_ * a.ldx;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = 16 * u + 4 * q;
            if (vec_ok && k + 3 < K1) {
                const float4 v = *reinterpret_cast<const float4*>(xr + k);
                xb_[u * 4 + 0] = v.x; xb_[u * 4 + 1] = v.y; xb_[u * 4 + 2] = v.z; xb_[u * 4 + 3] = v.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) xb_[u * 4 + r] = (k + r) < K1 ? xr[k + r] : 0.0f;
            }
        }
        if (!valid_) {
#pragma unroll
            for (int s = 0; s < KU * 4; ++s) xb_[s] = 0.0f;
        }
    };
    const int64_t tile_stride = (int64_t)gridDim.x * 4;
    // narrow inputs leave registers for a software pipeline: the next tile's fragment is fetched while this tile
    // computes.  The 140-wide critic already fills the register file (its prefetch spilled and ran 7 % slower).
    constexpr bool PREFETCH = KU <= 4;
    float xb[KU * 4];
    if constexpr (PREFETCH) {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + wave;
        if (t0 < n_tiles) load_x(t0, xb);
    }
    // STAGE: asynchronous copy of a tile into one half of this wavefront's double buffer.  Lanes past the tile's last
    // valid float4 re-read its first one (always valid) into slack or into rows that every consumer masks.
    // The copy is issued as inline assembly on purpose: told about an LDS-DMA (the builtin), the compiler cannot tell the two
    // halves of the double buffer apart and puts s_waitcnt vmcnt(0) in front of the next LDS read of ANY kind - the
    // "asynchronous" copy then cost a full HBM round trip at the top of every tile (3.7 k of a tile's 27 k cycles by
    // in-kernel cycle stamps).  Here the kernel waits for it itself, at the end of the tile (s_waitcnt(0) below).  M0 = LDS
    // byte address of the 1 KB slice (wave-uniform), lane l lands at M0 + 16 l.
    // By reference (a.x_rows): the 16 rows of a tile are scattered; lanes 0..15 hold their row numbers (rows16, fetched two
    // tiles ahead by tile_rows) and hand them to the copying lanes by shuffle: float4 number idx of the packed [16][K1] tile
    // comes from row idx / (K1/4), a multiply-high with kv_magic.
    const int kv = K1 >> 2;
    const uint32_t kv_magic = (uint32_t)((0x100000000ull + (uint32_t)(kv > 0 ? kv : 1) - 1u) / (uint32_t)(kv > 0 ? kv : 1));
    auto tile_rows = [&](int64_t tile_) {
        int v = 0;
        const int64_t r_ = tile_ * 16 + c;
        if (a.x_rows && q == 0 && tile_ < n_tiles && r_ < a.n_rows) v = a.x_rows[r_];
        return v;
    };
    auto stage = [&](int64_t tile_, float* dst, int rows16) {
        const float* gt = a.x + tile_ * 16 * (int64_t)K1;
        const int64_t rows_left = a.n_rows - tile_ * 16;
        const int nvv = (int)(rows_left < 16 ? rows_left : 16) * kv;
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)(dst);
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = lane + 64 * j;
            const int idv = idx < nvv ? idx : 0;
            const float* src = gt + 4 * idv;
            if (a.x_rows) {
                const int rr = (int)__umulhi((uint32_t)idv, kv_magic);
                src = a.x + (int64_t)__shfl(rows16, rr) * K1 + 4 * (idv - rr * kv);
            }
            const unsigned m0v = __builtin_amdgcn_readfirstlane(lds0 + 1024u * (unsigned)j);
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(m0v) : "memory", "m0");
        }
    };
    if constexpr (SPLIT) {
        if (grad_wave) {
            // ---- gradient wavefront: stages the pair's input tiles (one ahead of the activation wavefront) and accumulates
            //      dW1[j][i] += sum_b dZ1^T[j][b] X[b][i] and the layer-1 bias gradient, tile by tile -------------------------
            f32x4 gw[4][KU];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int u = 0; u < KU; ++u) gw[t][u] = zero4;
            float gb1 = 0.0f;
#ifdef EM_STAMP_BWD
            if (blockIdx.x == 1 && (threadIdx.x & 255) == 0) bwd_stamps[ACTOR][15] = 100 + __builtin_amdgcn_s_getreg(2308);   // SIMD id
#endif
            const float* const rrow = scr + c * TS + q;
            const float* const rsum = scr + lane * TS;
            const int64_t t0 = (int64_t)blockIdx.x * 4 + wave;
            float xv[4][KU];
            auto read_x = [&](int64_t tile_, const float* xt) {     // this lane's dW1 operands of a staged tile
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t rb = tile_ * 16 + 4 * s + q;
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const int i = 16 * u + c;
                        xv[s][u] = (rb < a.n_rows && i < K1) ? xt[(4 * s + q) * K1 + i] : 0.0f;
                    }
                }
            };
            if (t0 < n_tiles) stage(t0, xs, tile_rows(t0));
            int rows_nx = tile_rows(t0 + tile_stride);
            __builtin_amdgcn_s_waitcnt(0);
            post_flag(F_XREADY, 1);
            int k = 0, curg = 0;
            for (int64_t tile = t0; tile < n_tiles; tile += tile_stride, ++k, curg ^= 1) {
                // the other half held tile k - 1: the activation wavefront finished that tile (F_READY >= k was seen in the
                // previous iteration) and this wavefront's copies of its values were consumed by the products
                if (tile + tile_stride < n_tiles) stage(tile + tile_stride, xs + (curg ^ 1) * XBUF, rows_nx);
                rows_nx = tile_rows(tile + 2 * tile_stride);
                read_x(tile, xs + curg * XBUF);
                __builtin_amdgcn_s_waitcnt(0);            // the next tile has landed (and this tile's values are in registers)
                post_flag(F_XREADY, k + 2);
                wait_flag(F_READY, k + 1);                // the activation wavefront has finished tile k
                float tr_c[4][4];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s) tr_c[t][s] = rrow[16 * t * TS + 4 * s];
                float pb0 = 0.0f, pb1 = 0.0f;
#pragma unroll
                for (int j = 0; j < 16; j += 2) { pb0 += rsum[j]; pb1 += rsum[j + 1]; }
                gb1 += pb0 + pb1;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                post_flag(F_SDONE, k + 1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < KU; ++u)
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            gw[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_c[t][s], xv[s][u], gw[t][u], 0, 0, 0);
            }
            float* out = a.slab + ((int64_t)blockIdx.x * 4 + wave) * a.slab_stride;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int j = 16 * t + 4 * q + r;
#pragma unroll
                    for (int u = 0; u < KU; ++u) {
                        const int i = 16 * u + c;
                        if (i < K1) out[j * K1 + i] = poisoned ? __builtin_nanf("") : gw[t][u][r];
                    }
                }
            out[HID * K1 + lane] = gb1;
            return;
        }
    }
    // critic: a state's entry range {k, first} is fetched one tile ahead, so that its returns can be requested at the top of
    // the tile; they are consumed after the forward pass, which hides both HBM round trips (they were exposed: the loss read
    // ent_ret[first + j] right where it needed it, with a single wavefront per SIMD)
    constexpr int KPRE = 4;
    int n_k = 0;
    int n_first = 0;                                      // 32-bit on purpose: widening it here would wait for the load at once
    if constexpr (!ACTOR) {
        const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 16 + c;
        if (r0 < a.n_rows && q == 0) { n_k = a.st_k[r0]; n_first = a.st_first[r0]; }
    }
    int cur = 0;
    int kt = 0;                                           // SPLIT: tiles this pair has started
    int rows_next = 0;                                    // row numbers of the tile that the NEXT iteration stages
    if constexpr (STAGE && !SPLIT) {
        const int64_t t0 = (int64_t)blockIdx.x * 4 + wave;
        if (t0 < n_tiles) stage(t0, xs, tile_rows(t0));
        rows_next = tile_rows(t0 + tile_stride);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    unsigned long long bst_prev = 0; (void)bst_prev;
    unsigned bst_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; (void)bst_acc;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += tile_stride) {
        BSTAMP(0);
        const int64_t row = tile * 16 + c;
        const bool valid = row < a.n_rows;
        const float* xt = xs + cur * XBUF;            // STAGE: this tile, [16][K1] packed
        if constexpr (STAGE) {
            if constexpr (SPLIT) {
                wait_flag(F_XREADY, kt + 1);          // staged by the gradient wavefront, one tile ahead
                BSTAMP(9);
            } else {
                if (tile + tile_stride < n_tiles) stage(tile + tile_stride, xs + (cur ^ 1) * XBUF, rows_next);
                rows_next = tile_rows(tile + 2 * tile_stride);
            }
        } else if constexpr (!PREFETCH) load_x(tile, xb);
        // loss inputs of this tile: issued now, needed after the forward pass
        float p_act[4] = {0.f, 0.f, 0.f, 0.f}, p_adv = 0.0f, p_old = 0.0f;
        int p_k = 0;
        int64_t p_first = 0;
        float p_ret[KPRE] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ACTOR) {
            if (valid) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q + r < O) p_act[r] = a.action[row * O + 4 * q + r];
                p_adv = a.adv[row];
                p_old = a.old_logp[row];
            }
        } else {
            p_k = n_k; p_first = n_first;
            n_k = 0;
            const int64_t rn = row + tile_stride * 16;
            if (rn < a.n_rows && q == 0) { n_k = a.st_k[rn]; n_first = a.st_first[rn]; }
#pragma unroll
            for (int j = 0; j < KPRE; ++j) p_ret[j] = j < p_k ? a.ent_ret[p_first + j] : 0.0f;
        }
        BSTAMP(1);
        // ---- forward (recompute) -------------------------------------------------------------------
        f32x4 h1[4], h2[4], y;
        int boff = 4 * q;                             // opaque to the optimiser: the bias reads stay inside the loop
        asm volatile("" : "+v"(boff));
        // Weight fragments (A operands) come from LDS.  Every MFMA block below is software-pipelined by hand: the four
        // fragments aw[t] (tile t, 4 consecutive reduction indices) of the NEXT batch - or of the next block's first
        // batch - are requested in two halves, each right after the 8 MFMAs that consumed the registers it overwrites, i.e.
        // 8 MFMAs (256 cycles) before its first use; the four accumulators of a batch stay independent.
        // Left to itself the compiler reads a batch, waits for LDS, issues 16 MFMAs, and repeats: with one or two
        // wavefronts per SIMD that wait was exposed about 20 times per tile.  (Per accumulator the order of the partial
        // sums is unchanged, so the results are bitwise those of the plain loops.)
#define EM_SB() __builtin_amdgcn_sched_barrier(0)
        // one base register per access pattern; everything else is an immediate offset of the LDS instruction.  Row
        // strides are even, so a lane's four consecutive weights are two 8-byte reads.
        const float* rowW1 = W1s + c * S1 + boff;     // [feature row c][k = 4q ..]
        const float* rowW2 = rowW1 + (HID * S1 + c * (S2 - S1));
        const float* colW2 = W2s + boff * S2 + c;     // [k row 4q ..][feature column c]
        // half fragments: reduction indices r = 2h, 2h + 1 of tile t (h = 0, 1)
        auto ldRow = [&](const float* p, int h, f32x4& a) {
            const float2 v = *reinterpret_cast<const float2*>(p + 2 * h);
            a[2 * h] = v.x; a[2 * h + 1] = v.y;
        };
        auto ldCol = [&](const float* p, int h, f32x4& a) { a[2 * h] = p[2 * h * S2]; a[2 * h + 1] = p[(2 * h + 1) * S2]; };
        auto ldW1 = [&](int u, int t, int h, f32x4& a) { ldRow(rowW1 + 16 * t * S1 + 16 * u, h, a); };
        auto ldW2 = [&](int u, int t, int h, f32x4& a) { ldRow(rowW2 + 16 * t * S2 + 16 * u, h, a); };
        auto ldW3 = [&](int u, int h, f32x4& a) { ldRow(rowW2 + HID * S2 + 16 * u, h, a); };
        auto ldW3T = [&](int t, int h, f32x4& a) { ldCol(colW2 + HID * S2 + 16 * t, h, a); };
        auto ldW2T = [&](int u, int t, int h, f32x4& a) { ldCol(colW2 + 16 * u * S2 + 16 * t, h, a); };
        f32x4 aw[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            ldW1(0, t, 0, aw[t]); ldW1(0, t, 1, aw[t]);
            h1[t] = *reinterpret_cast<const f32x4*>(Bs + 16 * t + boff);
        }
        if constexpr (STAGE) {
            auto ldX = [&](int u) {
                const int k0 = 16 * u + 4 * q;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 < K1) xv = *reinterpret_cast<const float4*>(xt + c * K1 + k0);      // K1 % 4 == 0: whole or nothing
                return f32x4{xv.x, xv.y, xv.z, xv.w};
            };
            f32x4 xcur = ldX(0), xnext = xcur;
#pragma unroll
            for (int u = 0; u < KU; ++u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 2 * h; r < 2 * h + 2; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t) h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][r], xcur[r], h1[t], 0, 0, 0);
                    EM_SB();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (u + 1 < KU) ldW1(u + 1, t, h, aw[t]); else ldW2(0, t, h, aw[t]);
                    }
                    if (h == 0 && u + 1 < KU) xnext = ldX(u + 1);
                    EM_SB();
                }
                xcur = xnext;
            }
        } else {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int r = 2 * h; r < 2 * h + 2; ++r)
#pragma unroll
                        for (int t = 0; t < 4; ++t) h1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][r], xb[u * 4 + r], h1[t], 0, 0, 0);
                    EM_SB();
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (u + 1 < KU) ldW1(u + 1, t, h, aw[t]); else ldW2(0, t, h, aw[t]);
                    }
                    EM_SB();
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[t][r] = fmaxf(h1[t][r], 0.0f);
        // the x fragment is dead from here on: fetch the next tile's into the same registers (the rest of this
        // iteration hides the HBM latency; one wavefront per SIMD has nobody else to hide it)
        if constexpr (PREFETCH) {
#pragma unroll
            for (int sx = 0; sx < KU * 4; ++sx) xtr[(16 * (sx >> 2) + 4 * q + (sx & 3)) * TS + c] = xb[sx];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (tile + tile_stride < n_tiles) load_x(tile + tile_stride, xb);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) h2[t] = *reinterpret_cast<const f32x4*>(Bs + HID + 16 * t + boff);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 2 * h; r < 2 * h + 2; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) h2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][r], h1[u][r], h2[t], 0, 0, 0);
                EM_SB();
#pragma unroll
                for (int t = 0; t < 4; ++t) {      // last batch: layer 3's fragments, aw[t] = its batch t (critic: dZ1's first)
                    if (u < 3) ldW2(u + 1, t, h, aw[t]);
                    else if constexpr (ACTOR) ldW3(t, h, aw[t]);
                    else ldW2T(0, t, h, aw[t]);
                }
                EM_SB();
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) h2[t][r] = fmaxf(h2[t][r], 0.0f);
        y = bias3;
        // The critic's head has ONE output: as a 16-wide MFMA tile its forward, dH2 and dW3 products are 48 matrix
        // instructions of which 1/16 is used, a chain of 16 dependent MFMAs in the forward, and two transposes (dY, H2)
        // for dW3.  On the vector ALU the same three products are 16 multiply-adds each on the values the lane already
        // holds (h2[t][r]: feature 16t + 4q + r of row c): v = w3 . h2 by four partial sums and two lane exchanges,
        // dH2 = w3 * dv, and dW3 accumulates dv * h2 per lane (summed over the 16 rows' lanes when the slab row is written).
        f32x4 w3v[4];
        float y0 = 0.0f;
        if constexpr (ACTOR) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) y = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[u][r], h2[u][r], y, 0, 0, 0);
                EM_SB();
                ldW3T(u, 0, aw[u]); ldW3T(u, 1, aw[u]);   // the backward pass's first fragments, in flight during the loss
                EM_SB();
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) w3v[t] = *reinterpret_cast<const f32x4*>(W3s + 16 * t + boff);
            // normalised returns of the state's first KPRE entries (mappo_algorithm.py:312-313)
#pragma unroll
            for (int j = 0; j < KPRE; ++j) p_ret[j] = (p_ret[j] - a.norm_mean) / a.norm_scale;
            float ps[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                ps[t] = w3v[t][0] * h2[t][0];
#pragma unroll
                for (int r = 1; r < 4; ++r) ps[t] = fmaf(w3v[t][r], h2[t][r], ps[t]);
            }
            y0 = em_rows_sum((ps[0] + ps[1]) + (ps[2] + ps[3])) + b3_0;
        }

        BSTAMP(2);
        // ---- loss and dL/dy in accumulator layout: dy[r] = d/d out[row c][o = 4q + r] --------------
        f32x4 dy = zero4;
        if constexpr (ACTOR) {
            float d[4], lp = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * q + r;
                d[r] = 0.0f;
                if (o < O && valid) {
                    d[r] = p_act[r] - y[r];
                    lp += -(d[r] * d[r]) * (0.5f * ivar[r]) - ls[r] - 0.91893853320467274f;
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
                g_lp = -(A * ratio * wgt) * a.inv_L;                        // d(-mean(min))/d logp
                if (q == 0) { loss_sum += (double)fminf(su1, su2); cnt_sum += 1.0f; }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 4 * q + r;
                if (o < O && valid) {
                    const float iv = ivar[r];
                    dy[r] = g_lp * d[r] * iv;                               // d logp / d mu
                    gls[r] += g_lp * (d[r] * d[r] * iv - 1.0f) - a.ecoef * a.inv_L;   // d logp / d log_std ; entropy bonus
                }
            }
        } else {
            if (valid && q == 0) {
                const float v = y0;
                const int k = p_k;
                const int64_t first = p_first;
                float gsum = 0.0f;
                const float vn = (v - a.norm_mean) / a.norm_scale;
#pragma unroll
                for (int j = 0; j < KPRE; ++j) {
                    if (j < k) {
                        const float e = vn - p_ret[j];
                        gsum += e;
                        loss_sum += (double)(e * e);                                  // :314 before the 0.5 and the mean
                    }
                }
                for (int j = KPRE; j < k; ++j) {
                    const float e = vn - (a.ent_ret[first + j] - a.norm_mean) / a.norm_scale;
                    gsum += e;
                    loss_sum += (double)(e * e);
                }
                cnt_sum += (float)k;
                dy[0] = a.vcoef * a.inv_L * gsum / a.norm_scale;            // d(vcoef * 0.5 * mean(e^2)) / dv
            }
        }

        BSTAMP(3);
        // ---- backward through layer 3 and 2 ----------------------------------------------------------
        // The weight-gradient products below need their operands with the batch index on the reduction axis: one
        // 64 x 16 transpose per operand through the per-wavefront LDS scratch.  Done where they are needed, each transpose
        // is an exposed LDS round trip (write, read, wait) in front of an MFMA block, 5.5 of them per tile with one or two
        // wavefronts per SIMD.  Here the LDS instructions of a transpose are issued BETWEEN the MFMA batches of the block
        // before the one that consumes it, as soon as the source registers hold their final value:
        //   dY, H2 (for dW3)  under the 64 MFMAs of dZ1;   dZ2 (for dW2)  under the 16 MFMAs of dW3;
        //   dZ1 (for dW1)     under the 64 MFMAs of dW2.   Only H1 (for dW2) is still transposed in the open.
        // One scratch area is enough: the LDS executes a wavefront's instructions in order, so a later write cannot
        // overtake an earlier read.  The transposed copies take the registers their sources free (no higher peak).
        gb3 += dy;
        f32x4 dz2[4], dz1[4];
        if constexpr (ACTOR) {
#pragma unroll
            for (int t = 0; t < 4; ++t) dz2[t] = zero4;
#pragma unroll
            for (int h = 0; h < 2; ++h) {      // dH2^T = W3^T . dY^T, reduction over o = 4q + r
#pragma unroll
                for (int r = 2 * h; r < 2 * h + 2; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dz2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][r], dy[r], dz2[t], 0, 0, 0);
                EM_SB();
#pragma unroll
                for (int t = 0; t < 4; ++t) ldW2T(0, t, h, aw[t]);
                EM_SB();
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) dz2[t][r] = h2[t][r] > 0.0f ? dz2[t][r] : 0.0f;
        } else {
            const float dv = em_rows_sum(dy[0]);       // dL/dv of row c: only the lane with q = 0 evaluated the loss, the others hold 0
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gW3[t][r] = fmaf(dv, h2[t][r], gW3[t][r]);                    // dW3[0][16t + 4q + r], this lane's row
                    dz2[t][r] = h2[t][r] > 0.0f ? w3v[t][r] * dv : 0.0f;          // dH2 through the ReLU
                }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) dz1[t] = zero4;
        // transposed fragments: tr[t][s] = X[feature 16t + c][batch 4s + q]
        float tr_y[4], tr_a[4][4], tr_b[4][4], tr_c[4][4];
        float* const wcol = scr + 4 * q * TS + c;      // element (feature 16t + 4q + r, batch c) goes to wcol[(16t + r) * TS]
        const float* const rrow = scr + c * TS + q;    // tr[t][s] comes from rrow[16t * TS + 4s]
        const float* const rsum = scr + lane * TS;     // row `lane` of a transposed tile: feature `lane` over the 16 batch rows
        auto opB = [&](int i) {                         // 40 LDS instructions: dY and H2 out, tr_y and tr_b (for dW3) back
            if (i < 4) wcol[i * TS] = dy[i];
            else if (i < 8) tr_y[i - 4] = rrow[4 * (i - 4)];
            else if (i < 24) wcol[(16 * ((i - 8) >> 2) + ((i - 8) & 3)) * TS] = h2[(i - 8) >> 2][(i - 8) & 3];
            else tr_b[(i - 24) >> 2][(i - 24) & 3] = rrow[16 * ((i - 24) >> 2) * TS + 4 * ((i - 24) & 3)];
        };
        // critic (no dY / H2 transposes: its dW3 is on the vector ALU): everything dW2 needs goes through the scratch under
        // dZ1's MFMAs instead - dZ2 out, tr_a and the layer-2 bias row sums back, H1 out, tr_b back (80 LDS instructions)
        float pv0 = 0.0f, pv1 = 0.0f;
        auto opV = [&](int i) {
            if (i < 16) wcol[(16 * (i >> 2) + (i & 3)) * TS] = dz2[i >> 2][i & 3];
            else if (i < 32) tr_a[(i - 16) >> 2][(i - 16) & 3] = rrow[16 * ((i - 16) >> 2) * TS + 4 * ((i - 16) & 3)];
            else if (i < 48) { if ((i & 1) == 0) pv0 += rsum[i - 32]; else pv1 += rsum[i - 32]; }
            else if (i < 64) wcol[(16 * ((i - 48) >> 2) + ((i - 48) & 3)) * TS] = h1[(i - 48) >> 2][(i - 48) & 3];
            else tr_b[(i - 64) >> 2][(i - 64) & 3] = rrow[16 * ((i - 64) >> 2) * TS + 4 * ((i - 64) & 3)];
        };
        if constexpr (SPLIT) wait_flag(F_SDONE, kt);   // the scratch still holds the previous tile's dZ1^T until it is read
#pragma unroll
        for (int u = 0; u < 4; ++u) {      // dH1^T = W2^T . dZ2^T, reduction over k = 16u + 4q + r
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int r = 2 * h; r < 2 * h + 2; ++r)
#pragma unroll
                    for (int t = 0; t < 4; ++t) dz1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[t][r], dz2[u][r], dz1[t], 0, 0, 0);
                EM_SB();
                if (u < 3) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) ldW2T(u + 1, t, h, aw[t]);
                }
                if constexpr (ACTOR) {
#pragma unroll
                    for (int i = 0; i < 5; ++i) opB(5 * (2 * u + h) + i);
                } else {
#pragma unroll
                    for (int i = 0; i < 10; ++i) opV(10 * (2 * u + h) + i);
                }
                EM_SB();
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dz1[t][r] = h1[t][r] > 0.0f ? dz1[t][r] : 0.0f;
        }

        BSTAMP(4);
        // ---- weight gradients: reduction over the 16 batch rows of the tile ---------------------------
        // dW3[o][j] += sum_b dY^T[o][b] H2^T[j][b]; dZ2 leaves for the scratch between its MFMAs
        if constexpr (ACTOR)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                gW3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_y[s], tr_b[t][s], gW3[t], 0, 0, 0);
                wcol[(16 * s + t) * TS] = dz2[s][t];
            }
            EM_SB();
        }
        // rowsum: sum over the tile's 16 batch rows of feature `lane` (the bias gradient; rows past n_rows hold zeros)
        auto row_sum = [&](float& acc) {
            float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; j += 2) { p0 += rsum[j]; p1 += rsum[j + 1]; }
            acc += p0 + p1;
        };
        if constexpr (ACTOR) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) tr_a[t][s] = rrow[16 * t * TS + 4 * s];
        row_sum(gb2_l);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        BSTAMP(5);
        // H1 (its last use above was the ReLU mask of dZ1)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) wcol[(16 * t + r) * TS] = h1[t][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s) tr_b[t][s] = rrow[16 * t * TS + 4 * s];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        } else {
            gb2_l += pv0 + pv1;
            BSTAMP(5);
        }
        BSTAMP(6);
        // dW2[j2][j1] += sum_b dZ2^T[j2][b] H1^T[j1][b]; dZ1 is transposed (and row-summed: gb1) between the batches
        float pb0 = 0.0f, pb1 = 0.0f;
        auto opC = [&](int i) {                         // 48 LDS instructions: dZ1 out, tr_c back, 16 row-sum reads
            if (i < 16) wcol[(16 * (i >> 2) + (i & 3)) * TS] = dz1[i >> 2][i & 3];
            else if (i < 32) tr_c[(i - 16) >> 2][(i - 16) & 3] = rrow[16 * ((i - 16) >> 2) * TS + 4 * ((i - 16) & 3)];
            else if ((i & 1) == 0) pb0 += rsum[i - 32];
            else pb1 += rsum[i - 32];
        };
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                for (int t2 = 2 * hh; t2 < 2 * hh + 2; ++t2)
#pragma unroll
                    for (int t1 = 0; t1 < 4; ++t1)
                        gW2[t2][t1] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_a[t2][s], tr_b[t1][s], gW2[t2][t1], 0, 0, 0);
                EM_SB();
                if constexpr (SPLIT) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) opC(2 * (2 * s + hh) + i);      // dZ1 out only: the gradient wavefront reads it
                } else {
#pragma unroll
                    for (int i = 0; i < 6; ++i) opC(6 * (2 * s + hh) + i);
                }
                EM_SB();
            }
        }
        gb1_l += pb0 + pb1;
#undef EM_SB
        if constexpr (SPLIT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            post_flag(F_READY, kt + 1);
            ++kt;
        }
        // dW1[j][i] += sum_b dZ1^T[j][b] X[b][i]
        if constexpr (!SPLIT)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int64_t rb = tile * 16 + 4 * s + q;
            const float* xrow = xt + (4 * s + q) * K1;
            if constexpr (!STAGE && !PREFETCH) {          // wide inputs whose rows cannot be staged (K1 % 4 != 0)
                int64_t src_ = rb < a.n_rows ? rb : 0;
                if (a.x_rows) src_ = rb < a.n_rows ? (int64_t)a.x_rows[src_] : 0;   // states held by reference
                xrow = a.x + src_ * a.ldx;
            }
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const int i = 16 * u + c;
                float xv;
                if constexpr (PREFETCH) xv = xtr[i * TS + 4 * s + q];      // zero where row >= n_rows or i >= K1 (load_x)
                else xv = (rb < a.n_rows && i < K1) ? xrow[i] : 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    gW1[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(tr_c[t][s], xv, gW1[t][u], 0, 0, 0);
            }
        }
        BSTAMP(7);
        if constexpr (STAGE && !SPLIT) {              // the next tile has landed; this tile's half may be overwritten
            __builtin_amdgcn_s_waitcnt(0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (STAGE) cur ^= 1;
        BSTAMP(8);
    }

#ifdef EM_STAMP_BWD
    if (blockIdx.x == 1 && threadIdx.x == 0)
    {
        for (int i = 0; i < 10; ++i) bwd_stamps[ACTOR][i] += bst_acc[i];
        for (int i = 0; i < 4; ++i) bwd_stamps[ACTOR][10 + i] += spin_acc[i];
        bwd_stamps[ACTOR][14] = 100 + __builtin_amdgcn_s_getreg(2308);
    }
#endif
    // ---- write this wavefront's slab row: [w1 | b1 | w2 | b2 | w3 | b3 | (log_std) | loss, aux, count] ----
    float* out = a.slab + ((int64_t)blockIdx.x * 4 + wave) * a.slab_stride;
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
    if constexpr (ACTOR) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * q + r < O) out[o_w3 + (4 * q + r) * HID + 16 * t + c] = gW3[t][r];
    } else {
        // value head: gW3[t][r] is this lane's (row c) share of dW3[0][16t + 4q + r]; sum the 16 rows' lanes (fixed order)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = gW3[t][r];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) v += __shfl_xor(v, m);
                if (c == 0) out[o_w3 + 16 * t + 4 * q + r] = v;
            }
        for (int i = HID + lane; i < O * HID; i += 64) out[o_w3 + i] = 0.0f;    // only output 0 enters the value loss
    }
    // bias-type gradients: reduce the 16 lanes that share q (fixed butterfly order)
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            gb3[r] += __shfl_xor(gb3[r], m);
            gls[r] += __shfl_xor(gls[r], m);
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
        for (int r = 0; r < 4; ++r)
            if (4 * q + r < O) {
                out[o_b3 + 4 * q + r] = gb3[r];
                if constexpr (ACTOR) out[o_ls + 4 * q + r] = gls[r];
            }
    }
    if (lane == 0) { out[o_loss] = (float)loss_sum; out[o_loss + 1] = (float)((double)ent_row * (double)cnt_sum);   /* the entropy is the same for every row */ out[o_loss + 2] = cnt_sum; }
}

// grads[p] = sum over slab rows in a fixed order (bitwise reproducible): a workgroup owns 32 columns; its 8 row groups sum
// rows rg, rg + 8, ... into four interleaved partial sums each, and thread column p adds the 8 group sums in order.
// (One thread per column walking every row alone took 340 us for the critic's 13,249 columns x 1,024 rows.)
__global__ __launch_bounds__(256) void reduce_slabs_kernel(const float* __restrict__ slab, int64_t n_rows, int64_t stride,
                                                           int64_t n_cols, float* __restrict__ out, int64_t n_main,
                                                           float* __restrict__ out_tail)
{
    // columns [0, n_main) go to out, the columns behind them (the three loss sums) to out_tail: one launch for both
    __shared__ float part[8][33];
    const int col = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t p = (int64_t)blockIdx.x * 32 + col;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (p < n_cols) {
        const float* base = slab + p;
        int64_t r = rg;
        for (; r + 24 < n_rows; r += 32) {
            s0 += base[r * stride];
            s1 += base[(r + 8) * stride];
            s2 += base[(r + 16) * stride];
            s3 += base[(r + 24) * stride];
        }
        for (; r < n_rows; r += 8) s0 += base[r * stride];
    }
    part[rg][col] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0 && p < n_cols) {
        float s = part[0][col];
#pragma unroll
        for (int g = 1; g < 8; ++g) s += part[g][col];
        if (p < n_main) out[p] = s;
        else out_tail[p - n_main] = s;
    }
}

// clip_grad_norm_ over [0,n) followed by torch.optim.Adam's single-tensor update; one workgroup per parameter group
struct AdamGroup {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;
    float step_size, one_m_beta1, beta2, one_m_beta2, eps, max_norm;   // step_size = lr / (1 - beta1^step)
    float bias_c2_sqrt;             // sqrt(1 - beta2^step)
};

__global__ __launch_bounds__(1024) void adam_kernel(const AdamGroup g0, const AdamGroup g1, float* __restrict__ norms_out)
{
    const AdamGroup& g = blockIdx.x == 0 ? g0 : g1;
    __shared__ double red[1024];
    double ss = 0.0;
    for (int64_t i = threadIdx.x; i < g.n; i += 1024) ss += (double)g.grad[i] * (double)g.grad[i];
    red[threadIdx.x] = ss;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
        __syncthreads();
    }
    const float total_norm = (float)sqrt(red[0]);
    float coef = g.max_norm / (total_norm + 1e-6f);          // torch.nn.utils.clip_grad_norm_
    coef = coef > 1.0f ? 1.0f : coef;
    if (threadIdx.x == 0 && norms_out) norms_out[blockIdx.x] = total_norm;
    // A gradient that is not finite (a lost hand-over of the two-wavefront backward kernel poisons its slab with NaN; a
    // diverged run does the same) must not reach the parameters: the step is skipped for this network, the norm that the
    // host reads back says why, and MAPPO.update() raises.  (torch would write NaN into every parameter here.)
    if (!(total_norm <= 3.0e38f)) return;
    const float step_size = g.step_size;
    for (int64_t i = threadIdx.x; i < g.n; i += 1024) {
        const float gr = g.grad[i] * coef;
        const float m = g.exp_avg[i] + (gr - g.exp_avg[i]) * g.one_m_beta1;           // lerp_
        const float v = g.exp_avg_sq[i] * g.beta2 + g.one_m_beta2 * gr * gr;           // mul_().addcmul_()
        g.exp_avg[i] = m;
        g.exp_avg_sq[i] = v;
        const float denom = sqrtf(v) / g.bias_c2_sqrt + g.eps;
        g.param[i] = g.param[i] - step_size * (m / denom);                                // addcdiv_
    }
}

int check_net(const em_mlp3* net)
{
    if (!net || !net->w1 || !net->b1 || !net->w2 || !net->b2 || !net->w3 || !net->b3)
        return em::fail(EM_ERR_INVALID, "mlp3: null parameter pointer");
    if (net->hidden != HID || net->in_dim < 1 || net->in_dim > 144 || net->out_dim < 1 || net->out_dim > 16)
        return em::fail(EM_ERR_INVALID, "mlp3: unsupported shape (hidden 64, in_dim<=144, out_dim<=16)");
    return EM_OK;
}

template <bool ACTOR>
int launch_bwd(const BwdArgs& ba, int64_t n_waves, hipStream_t s)
{
    const int ku = (ba.net.in_dim + 15) / 16;
    const int s1 = ((ku * 16 + 2) | 2);
    // staged input tiles (see mlp3_bwd_kernel): wide inputs whose 16-row tiles are contiguous, aligned spans
    const bool stage = ku > 4 && ba.ldx == ba.net.in_dim && (ba.net.in_dim & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(ba.x) & 15) == 0;
    const size_t lds = sizeof(float) * ((size_t)HID * s1 + (size_t)HID * S2 + 16 * S2 + 2 * HID + 4 * HID * TS +
                                        (stage ? 4 * 2 * (size_t)((4 * ku * 16 + 63) / 64) * 256 : 0) +
                                        (ku <= 4 ? 4 * (size_t)ku * 16 * TS : 0));
    // two wavefronts per SIMD with the dW1 accumulators in the second one (SPLIT, see mlp3_bwd_kernel); EM_BWD_SPLIT=0 keeps
    // the one-wavefront kernel (tests compare the two bit for bit)
    const char* split_env = getenv("EM_BWD_SPLIT");
    const bool split = stage && !(split_env && split_env[0] == '0');
    const dim3 grid((unsigned)(n_waves / 4)), block(split ? 512 : 256);
    const size_t lds_all = lds + (split ? 64 : 0);
#define EM_BWD_ONE(KU_, STAGE_, SPLIT_)                                                                           \
    {                                                                                                             \
        if (lds_all > 64 * 1024) {                                                                                \
            hipError_t e = hipFuncSetAttribute((const void*)mlp3_bwd_kernel<KU_, ACTOR, STAGE_, SPLIT_>,           \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_all);         \
            if (e != hipSuccess) return em::hip_fail(e, "hipFuncSetAttribute");                                   \
        }                                                                                                         \
        hipLaunchKernelGGL((mlp3_bwd_kernel<KU_, ACTOR, STAGE_, SPLIT_>), grid, block, lds_all, s, ba);           \
    }
#define EM_BWD(KU_)                                                                                               \
    case KU_: {                                                                                                   \
        if constexpr (KU_ > 4) {                                                                                  \
            if (split) EM_BWD_ONE(KU_, true, true)                                                                \
            else if (stage) EM_BWD_ONE(KU_, true, false)                                                          \
            else EM_BWD_ONE(KU_, false, false)                                                                    \
        } else EM_BWD_ONE(KU_, false, false)                                                                      \
    } break
    switch (ku) {
        EM_BWD(1); EM_BWD(2); EM_BWD(3); EM_BWD(4); EM_BWD(5); EM_BWD(6); EM_BWD(7); EM_BWD(8); EM_BWD(9);
        default: return em::fail(EM_ERR_INVALID, "mlp3 backward: unsupported in_dim");
    }
#undef EM_BWD
#undef EM_BWD_ONE
    return em::hip_check(hipGetLastError(), "mlp3_bwd_kernel launch");
}

}  // namespace

extern "C" {
#ifdef EM_STAMP_BWD
int em_debug_bwd_stamps(unsigned long long* out, int reset)
{
    if (reset) { unsigned long long z[32] = {0}; return (int)hipMemcpyToSymbol(HIP_SYMBOL(bwd_stamps), z, sizeof z); }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(bwd_stamps), sizeof(unsigned long long) * 32);
}
#endif

int em_gae(const em_rollout* ro, double gamma, double lam, double* partials, int64_t partials_capacity, double* moments3,
           void* stream)
{
    if (!ro || !partials || !moments3) return em::fail(EM_ERR_INVALID, "em_gae: null argument");
    const int64_t blocks = (ro->n_envs + 255) / 256;
    if (partials_capacity < 3 * blocks) return em::fail(EM_ERR_INVALID, "em_gae: partials buffer needs %lld doubles", (long long)(3 * blocks));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ro->n_envs, ro->horizon, ro->reward_tb,
                       ro->value_tb, ro->done_tb, ro->k_tb, ro->first_tb, gamma, lam, ro->ent_adv, ro->ent_ret, partials);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, partials, blocks, moments3);
    return em::hip_check(hipGetLastError(), "gae launch");
}

int em_gae_static(const em_rollout* ro, double gamma, double lam, double* moments3, void* stream)
{
    if (!ro || !moments3) return em::fail(EM_ERR_INVALID, "em_gae_static: null argument");
    if (ro->n_envs < 1 || ro->n_envs > EM_GAE_STATIC_MAX_ENVS)
        return em::fail(EM_ERR_INVALID, "em_gae_static: n_envs %lld is outside 1..%lld (the library's fixed scratch); use em_gae "
                        "with a caller-owned partials buffer", (long long)ro->n_envs, (long long)EM_GAE_STATIC_MAX_ENVS);
    int dev = -1;
    EM_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= GAE_MAX_DEVICES) return em::fail(EM_ERR_INVALID, "em_gae_static: device index %d out of range", dev);
    // the scratch pointer belongs to the CURRENT device; a stream of another GPU would run the kernels there with this
    // GPU's address.  The precondition "current device == the stream's device" is checked, not assumed.
    hipDevice_t sdev = dev;
    if (stream && hipStreamGetDevice((hipStream_t)stream, &sdev) == hipSuccess && (int)sdev != dev)
        return em::fail(EM_ERR_INVALID, "em_gae_static: the stream belongs to device %d but device %d is current; make the "
                        "stream's device current (hipSetDevice) before the call", (int)sdev, dev);
    double* part = g_gae_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!part) {
        // once per device, with that device current: every GPU holds its own copy of the array at its own address
        EM_HIP(hipGetSymbolAddress((void**)&part, HIP_SYMBOL(scratch)));
        if (!part) return em::fail(EM_ERR_HIP, "em_gae_static: the scratch symbol has no device address");
        g_gae_scratch_ptr[dev].store(part, std::memory_order_release);
    }
    const int64_t blocks = (ro->n_envs + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    // rows [0, blocks) of the scratch are written in full by gae_kernel before sum_partials_kernel reads exactly those
    // rows, so whatever an earlier (larger) call left behind is never read and nothing has to be cleared
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ro->n_envs, ro->horizon, ro->reward_tb,
                       ro->value_tb, ro->done_tb, ro->k_tb, ro->first_tb, gamma, lam, ro->ent_adv, ro->ent_ret, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, part, blocks, moments3);
    return em::hip_check(hipGetLastError(), "gae (static scratch) launch");
}

int64_t em_ppo_slab_floats(int32_t in_dim, int32_t out_dim, int is_actor)
{
    return (int64_t)HID * in_dim + HID + HID * HID + HID + (int64_t)out_dim * HID + out_dim + (is_actor ? out_dim : 0) + 3;
}

int em_ppo_actor_grad(const em_mlp3* actor, const float* log_std, const float* ent_obs, const float* ent_action,
                      const float* ent_logp, const float* ent_adv, const double* adv_moments3, int64_t n_entries,
                      int64_t n_global, float clip, float entropy_coeff, float* slab, int64_t n_waves, float* grad_out,
                      float* loss3_out, void* stream)
{
    int rc = check_net(actor);
    if (rc) return rc;
    if (!log_std || !ent_obs || !ent_action || !ent_logp || !ent_adv || !adv_moments3 || !slab || !grad_out || !loss3_out)
        return em::fail(EM_ERR_INVALID, "em_ppo_actor_grad: null argument");
    if (n_entries < 1 || n_global < n_entries || n_waves < 4 || (n_waves & 3))
        return em::fail(EM_ERR_INVALID, "em_ppo_actor_grad: bad sizes");
    BwdArgs ba;
    std::memset(&ba, 0, sizeof ba);
    ba.net = *actor; ba.x = ent_obs; ba.ldx = actor->in_dim; ba.n_rows = n_entries; ba.slab = slab;
    ba.slab_stride = em_ppo_slab_floats(actor->in_dim, actor->out_dim, 1);
    ba.inv_L = 1.0f / (float)n_global;
    ba.log_std = log_std; ba.action = ent_action; ba.old_logp = ent_logp; ba.adv = ent_adv; ba.moments = adv_moments3;
    ba.clip = clip; ba.ecoef = entropy_coeff;
    hipStream_t s = (hipStream_t)stream;
    rc = launch_bwd<true>(ba, n_waves, s);
    if (rc) return rc;
    const int64_t np = ba.slab_stride - 3;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((np + 3 + 31) / 32)), dim3(256), 0, s, slab, n_waves, ba.slab_stride, np + 3, grad_out,
                       np, loss3_out);
    return em::hip_check(hipGetLastError(), "actor grad launch");
}

int em_ppo_critic_grad(const em_mlp3* critic, const float* st_state, const int32_t* st_first, const uint8_t* st_k,
                       const float* ent_ret, int64_t n_steps, int64_t n_global, float value_loss_coeff,
                       float value_norm_mean, float value_norm_var, float* slab, int64_t n_waves, float* grad_out,
                       float* loss3_out, void* stream)
{
    return em_ppo_critic_grad_rows(critic, st_state, nullptr, st_first, st_k, ent_ret, n_steps, n_global, value_loss_coeff,
                                   value_norm_mean, value_norm_var, slab, n_waves, grad_out, loss3_out, stream);
}

int em_ppo_critic_grad_rows(const em_mlp3* critic, const float* state_base, const int32_t* st_row, const int32_t* st_first,
                            const uint8_t* st_k, const float* ent_ret, int64_t n_steps, int64_t n_global,
                            float value_loss_coeff, float value_norm_mean, float value_norm_var, float* slab, int64_t n_waves,
                            float* grad_out, float* loss3_out, void* stream)
{
    int rc = check_net(critic);
    if (rc) return rc;
    if (!state_base || !st_first || !st_k || !ent_ret || !slab || !grad_out || !loss3_out)
        return em::fail(EM_ERR_INVALID, "em_ppo_critic_grad: null argument");
    if (n_steps < 1 || n_global < 1 || n_waves < 4 || (n_waves & 3)) return em::fail(EM_ERR_INVALID, "em_ppo_critic_grad: bad sizes");
    BwdArgs ba;
    std::memset(&ba, 0, sizeof ba);
    ba.net = *critic; ba.x = state_base; ba.x_rows = st_row; ba.ldx = critic->in_dim; ba.n_rows = n_steps; ba.slab = slab;
    ba.slab_stride = em_ppo_slab_floats(critic->in_dim, critic->out_dim, 0);
    ba.inv_L = 1.0f / (float)n_global;
    ba.st_first = st_first; ba.st_k = st_k; ba.ent_ret = ent_ret;
    ba.vcoef = value_loss_coeff; ba.norm_mean = value_norm_mean; ba.norm_scale = sqrtf(value_norm_var + 1e-8f);
    hipStream_t s = (hipStream_t)stream;
    rc = launch_bwd<false>(ba, n_waves, s);
    if (rc) return rc;
    const int64_t np = ba.slab_stride - 3;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)((np + 3 + 31) / 32)), dim3(256), 0, s, slab, n_waves, ba.slab_stride, np + 3, grad_out,
                       np, loss3_out);
    return em::hip_check(hipGetLastError(), "critic grad launch");
}

int em_adam_step(float* actor_param, const float* actor_grad, float* actor_m, float* actor_v, int64_t n_actor, double lr_actor,
                 float* critic_param, const float* critic_grad, float* critic_m, float* critic_v, int64_t n_critic,
                 double lr_critic, int64_t step, double beta1, double beta2, double eps, double max_grad_norm, float* norms2_out,
                 void* stream)
{
    if (!actor_param || !actor_grad || !actor_m || !actor_v || !critic_param || !critic_grad || !critic_m || !critic_v)
        return em::fail(EM_ERR_INVALID, "em_adam_step: null argument");
    if (step < 1) return em::fail(EM_ERR_INVALID, "em_adam_step: step counts from 1");
    AdamGroup g[2];
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    const float omb1 = (float)(1.0 - beta1), omb2 = (float)(1.0 - beta2);
    g[0] = {actor_param, actor_grad, actor_m, actor_v, n_actor, (float)((double)lr_actor / bc1), omb1, (float)beta2, omb2, (float)eps, (float)max_grad_norm, (float)sqrt(bc2)};
    g[1] = {critic_param, critic_grad, critic_m, critic_v, n_critic, (float)((double)lr_critic / bc1), omb1, (float)beta2, omb2, (float)eps, (float)max_grad_norm, (float)sqrt(bc2)};
    hipLaunchKernelGGL(adam_kernel, dim3(2), dim3(1024), 0, (hipStream_t)stream, g[0], g[1], norms2_out);
    return em::hip_check(hipGetLastError(), "adam launch");
}

}  // extern "C"
