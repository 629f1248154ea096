// edge_env.hip - batched EdgeSimGym reset/step for MI355X (gfx950), plus its C ABI.
//
// Replaces, for n_envs environments per launch, EdgeSimGym.reset / step / _generate_tasks_and_obs /
// _get_obs / _calculate_reward / _maybe_apply_env_exploration of the reference (MRRL.PY:188-521).
//
// Mapping: an environment owns a group of LPE lanes (16, 32 or 64) of a wavefront, agents/servers
// on lanes; a 256-thread workgroup holds 4 wavefronts.  Each wavefront stages the state records of
// its environments HBM -> LDS with 16-byte lane-linear loads, works on the LDS copy (the placement
// chain of MRRL.PY:397-438 is order dependent: agent k sees the load agent k-1 added), builds the
// dense observation tile in LDS and writes record and tile back with 16-byte lane-linear stores.
// All environment arithmetic is float64 in the reference's evaluation order; this file must be
// compiled with -ffp-contract=off (a fused multiply-add would change deadline decisions).
//
// Random numbers: TAPE=true consumes pre-drawn numbers (parity with the reference's MT19937
// stream); TAPE=false uses Philox4x32-10 addressed by (seed; env, episode, step, purpose, index)
// and the polynomial log/exp/sin below, which oracle/edge_oracle.c mirrors bit for bit.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "edgemarl.h"
#include "em_common.h"

#pragma clang fp contract(off)

// Optional in-kernel cycle stamps of workgroup 0 / lane 0 (tools/stamp_env.hip builds this file with
// -DEM_STAMP); compiled out of the library.
#ifdef EM_STAMP
__device__ unsigned long long em_stamps[16];
#define STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) em_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

namespace {

// ------------------------------------------------------------------------------------------------
// state record (one per environment, contiguous in HBM, 16-byte multiple)
//   f64 load[n] | f64 wl[n] | f64 data[n] | f64 deadline[n] | f64 sim_time, elapsed, sum_lat, sum_energy
//   i32 queue[n] | i32 step, n_tasks, n_viol, n_completed, n_pending, n_logged | i8 rank[n] | pad
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int rec_off_scal(int n) { return 32 * n; }
__host__ __device__ inline int rec_off_queue(int n) { return 32 * n + 32; }
__host__ __device__ inline int rec_off_int(int n) { return 36 * n + 32; }
__host__ __device__ inline int rec_off_rank(int n) { return 36 * n + 56; }
__host__ __device__ inline int rec_bytes(int n) { return (37 * n + 56 + 15) & ~15; }
enum { I_STEP = 0, I_TASKS, I_VIOL, I_DONE, I_PENDING, I_LOGGED };
enum { S_TIME = 0, S_ELAPSED, S_SUMLAT, S_SUMEN };

// Per-launch scalars (kernel argument, SGPRs).
struct EnvParams {
    int32_t n, horizon, rate_mode, explore;
    int32_t rec_bytes, obs_floats, wave_bytes, pad0;
    double eps;
    uint32_t seed_lo, seed_hi, episode, env_offset;
    int64_t n_envs, lat_cap;
};

// Per-handle constants, resident in HBM (read with scalar loads where they are used, which keeps them
// out of the kernel's long-lived SGPRs).  inv_* are correctly rounded reciprocals computed on the host:
// a quotient by one of these constants is formed as q0 = x*inv; q = fma(fma(-q0, d, x), inv, q0), the
// final steps of the hardware's own float64 division sequence (v_div_fmas after the residual), which is
// correctly rounded for the value ranges of the simulator (tests/test_env_gpu.py::test_const_division).
struct EnvConst {
    double rate_lo, rate_span, rate_mid, cycle, inv_cycle;
    double wl_lo, wl_span, data_lo, data_span, ddl_lo, ddl_span, bw_lo, bw_span, lat_lo, lat_span;
    double w_lat, w_en, w_viol, dt_min, dt_max, window0;
    double cap[EM_MAX_NODES], inv_cap[EM_MAX_NODES], inv_k[EM_MAX_NODES + 1];
};

__device__ inline double qdiv(double x, double d, double inv)
{
    const double q0 = x * inv;
    return __builtin_fma(__builtin_fma(-q0, d, x), inv, q0);
}

// fmod(t, c) for t >= 0, c > 0: the remainder is exactly representable, so one fused residual is exact.
__device__ inline double exact_fmod_pos(double t, double c, double inv_c)
{
    const double q = floor(t * inv_c);
    double r = __builtin_fma(-q, c, t);
    if (r < 0.0) r += c;
    else if (r >= c) r -= c;
    return r;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11) and deterministic float64 helpers
// ------------------------------------------------------------------------------------------------
enum { P_DT = 0, P_POIS, P_TASK_A, P_TASK_B, P_EXPL, P_EXPL_RES, P_NET };

__device__ inline uint4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per half round (v_mad_u64_u32) instead of separate hi/lo multiplies
        const uint64_t m0 = (uint64_t)0xD2511F53u * c0, m1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(m0 >> 32), lo0 = (uint32_t)m0;
        const uint32_t hi1 = (uint32_t)(m1 >> 32), lo1 = (uint32_t)m1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ inline double u53(uint32_t a, uint32_t b)
{
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

__device__ inline double det_log(double x)
{
    int e;
    double m = frexp(x, &e);
    if (m < 0.7071067811865476) { m = m * 2.0; e -= 1; }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 1.0 / 25.0;
#pragma unroll
    for (int den = 23; den >= 3; den -= 2) p = p * z + 1.0 / (double)den;
    p = p * z + 1.0;
    return (double)e * 0.6931471805599453 + 2.0 * s * p;
}

__device__ constexpr double inv_fact(int i)
{
    double f = 1.0;
    for (int k = 2; k <= i; ++k) f *= (double)k;
    return 1.0 / f;
}

__device__ inline double det_exp(double x)
{
    const double k = floor(x * 1.4426950408889634 + 0.5);
    const double r = x - k * 0.6931471805599453;
    double p = inv_fact(14);
#pragma unroll
    for (int i = 13; i >= 0; --i) p = p * r + inv_fact(i);
    return ldexp(p, (int)k);
}

__device__ inline double det_sin2pi(double phase)
{
    const double j = floor(phase * 4.0 + 0.5);
    const double x = 6.283185307179586 * (phase - j * 0.25);
    const double z = x * x;
    const int q = (int)j & 3;
    double v;
    if (q & 1) {
        double p = inv_fact(16);
        p = p * z + -inv_fact(14); p = p * z + inv_fact(12); p = p * z + -inv_fact(10); p = p * z + inv_fact(8);
        p = p * z + -inv_fact(6);  p = p * z + inv_fact(4);  p = p * z + -inv_fact(2);  p = p * z + inv_fact(0);
        v = p;
    } else {
        double p = -inv_fact(15);
        p = p * z + inv_fact(13); p = p * z + -inv_fact(11); p = p * z + inv_fact(9); p = p * z + -inv_fact(7);
        p = p * z + inv_fact(5);  p = p * z + -inv_fact(3);  p = p * z + inv_fact(1);
        v = p * x;
    }
    return (q & 2) ? -v : v;
}

// numpy's float add.reduce order for n <= 128 (np.mean of MRRL.PY:334-335 inherits it).
__device__ inline double np_sum_small(const double* a, int n)
{
    if (n < 8) {
        double r = -0.0;
        for (int i = 0; i < n; ++i) r += a[i];
        return 0.0 + r;
    }
    double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 += a[i]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
        r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res += a[i];
    return 0.0 + res;
}

// Lane-parallel pre-draw of the environment-level random numbers: lane 0 of an environment's group
// evaluates the P_DT block, lanes 1..3 the first three P_POIS blocks (six uniforms of the Knuth
// loop), lanes 4..9 P_TASK_A and lanes 10..15 P_TASK_B of the first six arrivals.  Later Poisson
// uniforms / arrivals (rare) are evaluated inline by the group.  The stream is counter addressed,
// so this is bit-identical to drawing them one after the other (oracle/edge_oracle.c does that).
enum { PRE_LANES = 16, PRE_POIS_U = 6, PRE_TASKS = 6 };

template <int LPE>
__device__ inline uint64_t group_bits(uint64_t ballot, int g)
{
    if constexpr (LPE == 64) return ballot;
    else return (ballot >> (g * LPE)) & ((1ull << LPE) - 1ull);
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <int LPE, bool TAPE, bool RESET>
__global__ __launch_bounds__(256, 6) void env_kernel(const EnvParams p, unsigned char* __restrict__ state,
                                                  const EnvConst* __restrict__ cc,
                                                  const int32_t* __restrict__ placement,
                                                  const float* __restrict__ resource, const em_tape tape,
                                                  const em_step_out out, double* __restrict__ lat_log,
                                                  int32_t* __restrict__ err_flag)
{
    constexpr int EPW = 64 / LPE;   // environments per wavefront
    extern __shared__ __align__(16) unsigned char smem[];

    const int n = p.n, w = n + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane / LPE, sl = lane % LPE;
    const int64_t env0 = ((int64_t)blockIdx.x * 4 + wave) * EPW;   // first environment of this wavefront
    if (env0 >= p.n_envs) return;                                     // whole wavefront idle (no barriers below)
    const int n_valid = (int)((p.n_envs - env0) < EPW ? (p.n_envs - env0) : EPW);
    const int64_t env = env0 + g;
    const bool valid = g < n_valid;
    const bool agent = valid && sl < n;

    // LDS carve-up for this wavefront
    unsigned char* wbase = smem + wave * p.wave_bytes;
    unsigned char* rec = wbase + g * p.rec_bytes;
    double* r_load = reinterpret_cast<double*>(rec);
    double* r_wl = r_load + n;
    double* r_data = r_wl + n;
    double* r_ddl = r_data + n;
    double* r_scal = r_ddl + n;
    int32_t* r_queue = reinterpret_cast<int32_t*>(rec + rec_off_queue(n));
    int32_t* r_int = reinterpret_cast<int32_t*>(rec + rec_off_int(n));
    int8_t* r_rank = reinterpret_cast<int8_t*>(rec + rec_off_rank(n));
    int off = EPW * p.rec_bytes;
    float* s_obs_w = reinterpret_cast<float*>(wbase + off);
    float* s_obs = s_obs_w + g * p.obs_floats;
    off += (EPW * p.obs_floats * 4 + 15) & ~15;
    double* s_cap = reinterpret_cast<double*>(wbase + off);      // capacity, 1/capacity, 1/k tables
    double* s_icap = s_cap + n;
    double* s_ik = s_icap + n;
    off += (3 * n + 1) * 8;
    double* s_lat = reinterpret_cast<double*>(wbase + off) + g * n;
    off += EPW * n * 8;
    double* s_en = reinterpret_cast<double*>(wbase + off) + g * n;
    off += EPW * n * 8;
    float* s_ratio = reinterpret_cast<float*>(wbase + off) + g * n;
    off += (EPW * n * 4 + 7) & ~7;
    // environment-level draws of the step, produced by one Philox block per lane (philox mode only)
    double* s_ra = reinterpret_cast<double*>(wbase + off) + g * PRE_LANES;
    off += EPW * PRE_LANES * 8;
    double* s_rb = reinterpret_cast<double*>(wbase + off) + g * PRE_LANES;
    off += EPW * PRE_LANES * 8;
    int32_t* s_rz = reinterpret_cast<int32_t*>(wbase + off) + g * PRE_LANES;

    // the step's actions do not depend on the staged record: issue their loads first
    int tgt_in = -1;
    float res_in = 1.0f;
    if constexpr (!RESET) {
        if (agent) { tgt_in = placement[env * n + sl]; res_in = resource[env * n + sl]; }
    }

    STAMP(0);
    // ---- stage in: records (or zeros at reset), zeroed observation tile, capacity table ---------
    {
        uint4* dst = reinterpret_cast<uint4*>(wbase);
        const int nvec = n_valid * p.rec_bytes / 16;
        if constexpr (RESET) {
            for (int i = lane; i < EPW * p.rec_bytes / 16; i += 64) dst[i] = make_uint4(0, 0, 0, 0);
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(state + env0 * p.rec_bytes);
            for (int i = lane; i < nvec; i += 64) dst[i] = src[i];
        }
        uint4* oz = reinterpret_cast<uint4*>(s_obs_w);
        for (int i = lane; i < ((EPW * p.obs_floats * 4 + 15) >> 4); i += 64) oz[i] = make_uint4(0, 0, 0, 0);
        for (int i = lane; i < n; i += 64) { s_cap[i] = cc->cap[i]; s_icap[i] = cc->inv_cap[i]; }
        for (int i = lane; i <= n; i += 64) s_ik[i] = cc->inv_k[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const double cap_me = agent ? s_cap[sl] : 1.0;
    const double icap_me = agent ? s_icap[sl] : 1.0;
    const uint32_t envid = p.env_offset + (uint32_t)env;
    const int step_no = RESET ? 0 : r_int[I_STEP] + 1;
    const double t_now = RESET ? 0.0 : r_scal[S_TIME];
    int draws_lane = 0;       // draws made on behalf of this lane's agent
    int draws_env = 0;        // draws made once per environment (uniform in the group)
    double reward = 0.0;

    if constexpr (!RESET) {
        STAMP(1);
        // ---- placement phase: per-agent quantities in parallel on the lanes (MRRL.PY:397-423) ----
        int rank = agent ? (int)r_rank[sl] : -1;
        int tgt = tgt_in;
        bool acting = rank >= 0 && tgt >= 0 && tgt < n;
        float res = acting ? res_in : 1.0f;
        const double wl = acting ? r_wl[sl] : 1.0;
        const double data = acting ? r_data[sl] : 1.0;
        const double ddl = acting ? r_ddl[sl] : 0.0;

        int slot = rank;   // acting slot = number of acting agents that arrived earlier
        if constexpr (TAPE) {
            const int key = acting ? rank : 127;
            slot = 0;
            for (int j = 0; j < n; ++j) {
                const int other = __shfl(key, g * LPE + j);
                slot += (other < key) ? 1 : 0;
            }
        }

        if (p.explore && acting) {   // _maybe_apply_env_exploration, MRRL.PY:258-274
            double u;
            uint32_t tgt_bits = 0;
            if constexpr (TAPE) u = tape.explore_u[env * n + slot];
            else {
                const uint4 b = philox(envid, p.episode, (uint32_t)step_no, (P_EXPL << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                u = u53(b.x, b.y);
                tgt_bits = b.z;
            }
            draws_lane += 1;
            if (!(u > p.eps)) {
                double r64;
                if constexpr (TAPE) {
                    tgt = tape.explore_tgt[env * n + slot];
                    r64 = tape.explore_res[env * n + slot];
                    if (tgt < 0 || tgt >= n) { atomicOr(err_flag, 1); tgt = 0; }
                } else {
                    tgt = (int)__umulhi(tgt_bits, (uint32_t)n);
                    const uint4 b2 = philox(envid, p.episode, (uint32_t)step_no, (P_EXPL_RES << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                    r64 = 0.1 + (1.0 - 0.1) * u53(b2.x, b2.y);
                }
                res = (float)r64;
                draws_lane += 2;
            }
        }

        double t_trans = 0.0;
        if (acting && tgt != sl) {   // MRRL.PY:409-414
            double bw_u, nlat;
            if constexpr (TAPE) { bw_u = tape.bw_u[env * n + slot]; nlat = tape.net_lat[env * n + slot]; }
            else {
                const uint4 b = philox(envid, p.episode, (uint32_t)step_no, (P_NET << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                bw_u = cc->bw_lo + cc->bw_span * u53(b.x, b.y);
                nlat = cc->lat_lo + cc->lat_span * u53(b.z, b.w);
            }
            draws_lane += 2;
            const double bw = bw_u * 1e6 / 8;
            t_trans = (data / bw) + nlat;
        }
        const double cap_t = acting ? s_cap[tgt] : 1.0;
        const double icap_t = acting ? s_icap[tgt] : 1.0;
        const double alloc = cap_t * (double)res;        // MRRL.PY:418 (float64 * float32)
        const double gcyc = qdiv(alloc, 1e9, 1e-9);
        const double t_exec = wl / gcyc;                  // MRRL.PY:419
        const double energy = wl * gcyc;                  // MRRL.PY:423
        const double inc = t_exec * gcyc;                 // MRRL.PY:437

        STAMP(2);
        // ---- ordered chain over arrival rank: queueing delay sees earlier placements (MRRL.PY:416,437)
        int k = 0;
        bool viol = false;
        for (int r = 0; r < n; ++r) {
            if (__ballot(acting && rank >= r) == 0ull) break;
            const bool mine = acting && rank == r;
            const uint64_t bal = __ballot(mine);
            if (mine) {
                const double l = r_load[tgt];
                const double t_queue = qdiv(l, cap_t, icap_t);
                const double t_total = t_trans + t_queue + t_exec;
                viol = (t_now + t_total) > ddl;          // MRRL.PY:431
                s_lat[k] = t_total;
                s_en[k] = energy;
                r_load[tgt] = l + inc;
                r_queue[tgt] += 1;
            }
            k += group_bits<LPE>(bal, g) != 0ull ? 1 : 0;
            __builtin_amdgcn_wave_barrier();
        }
        const int n_viol = __popcll(group_bits<LPE>(__ballot(viol), g));
        __builtin_amdgcn_wave_barrier();

        STAMP(3);
        // ---- reward (MRRL.PY:326-338) and episode accumulators (MRRL.PY:463-466) ----------------
        if (k > 0) {
            const double kd = (double)k, ik = s_ik[k];
            const double cost = (cc->w_lat * qdiv(np_sum_small(s_lat, k), kd, ik)) +
                                (cc->w_en * qdiv(np_sum_small(s_en, k), kd, ik)) + (cc->w_viol * (double)n_viol);
            reward = -cost;
        }
        const int tasks_before = r_int[I_TASKS];
        if (valid && sl == 0) {
            double sl_acc = r_scal[S_SUMLAT], se_acc = r_scal[S_SUMEN];
            for (int i = 0; i < k; ++i) { sl_acc += s_lat[i]; se_acc += s_en[i]; }
            r_scal[S_SUMLAT] = sl_acc;
            r_scal[S_SUMEN] = se_acc;
            r_int[I_TASKS] = tasks_before + k;
            r_int[I_VIOL] += n_viol;
            r_int[I_DONE] += k - n_viol;
        }
        if (agent) {
            if (out.step_latency) out.step_latency[env * n + sl] = sl < k ? s_lat[sl] : __builtin_nan("");
            if (lat_log && sl < k) {
                const int64_t at = (int64_t)tasks_before + sl;
                if (at < p.lat_cap) lat_log[env * p.lat_cap + at] = s_lat[sl];
            }
        }
        if (lat_log && valid && sl == 0) {
            const int64_t end = (int64_t)tasks_before + k;
            r_int[I_LOGGED] = (int32_t)(end < p.lat_cap ? end : p.lat_cap);
        }
        __builtin_amdgcn_wave_barrier();
    }

    STAMP(4);
    // ---- arrivals (MRRL.PY:340-383) ----------------------------------------------------------------
    if (agent) {
        r_rank[sl] = -1;
        s_ratio[sl] = (float)qdiv(r_load[sl], cap_me, icap_me);   // load/cap BEFORE this step's decay
    }
    double rate_u = 0.0, dt_u = 0.0;
    if constexpr (!TAPE) {
        if (sl < PRE_LANES) {
            const bool is_dt = sl == 0, is_pois = sl >= 1 && sl < 4, is_a = sl >= 4 && sl < 10;
            const uint32_t purpose = is_dt ? P_DT : (is_pois ? P_POIS : (is_a ? P_TASK_A : P_TASK_B));
            const uint32_t index = is_dt ? 0u : (uint32_t)(is_pois ? sl - 1 : (is_a ? sl - 4 : sl - 10));
            const uint4 b = philox(envid, p.episode, (uint32_t)step_no, (purpose << 24) | index, p.seed_lo, p.seed_hi);
            // value = lo + span * u; (lo, span) = (0, 1) leaves a raw uniform unchanged, bit for bit
            const double lo_a = is_dt || is_pois ? 0.0 : (is_a ? cc->wl_lo : cc->ddl_lo);
            const double sp_a = is_dt || is_pois ? 1.0 : (is_a ? cc->wl_span : cc->ddl_span);
            const double lo_b = is_dt ? cc->rate_lo : (is_a ? cc->data_lo : 0.0);
            const double sp_b = is_dt ? cc->rate_span : (is_a ? cc->data_span : 1.0);
            s_ra[sl] = lo_a + sp_a * u53(b.x, b.y);
            s_rb[sl] = lo_b + sp_b * u53(b.z, b.w);
            s_rz[sl] = (int32_t)__umulhi(b.z, (uint32_t)n);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        dt_u = s_ra[0];
        rate_u = s_rb[0];
    }
    STAMP(5);
    double rate;
    if (p.rate_mode == 3) rate = cc->rate_lo;      // degenerate range (MRRL.PY:222-223)
    else if (p.rate_mode == 1) {
        const double cyc = cc->cycle, icyc = cc->inv_cycle;   // cycle already clamped to >= 1e-6 (MRRL.PY:226)
        const double phase = qdiv(exact_fmod_pos(t_now, cyc, icyc), cyc, icyc);
        // tape mode never uses the rate (dt and the arrival count come pre-drawn)
        rate = cc->rate_lo + (0.5 + 0.5 * det_sin2pi(phase)) * cc->rate_span;
    } else if (p.rate_mode == 2) {
        if constexpr (TAPE) rate_u = valid ? tape.rate_u[env] : 0.0;
        rate = rate_u;
        draws_env += 1;
    } else rate = cc->rate_mid;
    const double rate_c = rate > 1e-6 ? rate : 1e-6;

    STAMP(6);
    double window;
    if constexpr (RESET) window = cc->window0;
    else {
        double raw;
        if constexpr (TAPE) raw = valid ? tape.dt_raw[env] : 0.0;
        else raw = (1.0 / rate_c) * (-det_log(1.0 - dt_u));
        draws_env += 1;
        const double dt_min = cc->dt_min, dt_max = cc->dt_max;
        window = raw < dt_min ? dt_min : (raw > dt_max ? dt_max : raw);   // MRRL.PY:247
    }

    STAMP(7);
    int n_new = 0;
    if constexpr (TAPE) {
        n_new = valid ? tape.n_arrivals[env] : 0;
        if (n_new > tape.arrival_slots || n_new < 0) { atomicOr(err_flag, 2); n_new = 0; }
    } else if (valid) {
        const double limit = det_exp(-(rate_c * window));
        double prod = 1.0;
        bool going = true;
#pragma unroll
        for (int i = 0; i < PRE_POIS_U; ++i) {   // Knuth multiplication method on the pre-drawn uniforms
            const double u = (i & 1) ? s_rb[1 + (i >> 1)] : s_ra[1 + (i >> 1)];
            prod = prod * u;
            going = going && prod > limit;
            n_new += going ? 1 : 0;
        }
        if (going) {
            uint4 b = make_uint4(0, 0, 0, 0);
            for (uint32_t i = PRE_POIS_U;; ++i) {   // rare tail; ends at n_new == 4096 at the latest
                if ((i & 1u) == 0u) b = philox(envid, p.episode, (uint32_t)step_no, (P_POIS << 24) | (i >> 1), p.seed_lo, p.seed_hi);
                const double u = (i & 1u) ? u53(b.z, b.w) : u53(b.x, b.y);
                prod = prod * u;
                if (prod > limit && n_new < 4096) ++n_new; else break;
            }
        }
    }
    draws_env += 1;
    __builtin_amdgcn_wave_barrier();

    STAMP(8);
    int n_pending = 0;
    for (int j = 0;; ++j) {
        const bool more = j < n_new;
        if (__ballot(more) == 0ull) break;
        if (more) {
            double wl, data, offs;
            int dec;
            if constexpr (TAPE) {
                const int64_t o = env * tape.arrival_slots + j;
                wl = tape.task_workload[o]; data = tape.task_data[o];
                offs = tape.task_deadline_offset[o]; dec = tape.task_decider[o];
                if (dec < 0 || dec >= n) { atomicOr(err_flag, 4); dec = 0; }
            } else if (j < PRE_TASKS) {
                wl = s_ra[4 + j]; data = s_rb[4 + j]; offs = s_ra[10 + j]; dec = s_rz[10 + j];
            } else {
                const uint4 a = philox(envid, p.episode, (uint32_t)step_no, (P_TASK_A << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                const uint4 b = philox(envid, p.episode, (uint32_t)step_no, (P_TASK_B << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                wl = cc->wl_lo + cc->wl_span * u53(a.x, a.y);
                data = cc->data_lo + cc->data_span * u53(a.z, a.w);
                offs = cc->ddl_lo + cc->ddl_span * u53(b.x, b.y);
                dec = (int)__umulhi(b.z, (uint32_t)n);
            }
            draws_env += 4;
            const double deadline = t_now + offs;          // MRRL.PY:217
            const bool is_new = r_rank[dec] < 0;           // broadcast read, before any lane updates
            const float q_now = (float)r_queue[dec];
            if (sl == 0) {
                if (is_new) {                               // MRRL.PY:361
                    r_rank[dec] = (int8_t)n_pending;
                    r_wl[dec] = wl; r_data[dec] = data; r_ddl[dec] = deadline;
                } else r_queue[dec] += 1;                  // MRRL.PY:359
            }
            if (is_new) {                                   // _get_obs, MRRL.PY:307-324
                const double rem = deadline - t_now;
                for (int c = sl; c < w; c += LPE) {
                    float v;
                    if (c == 0) v = s_ratio[dec];
                    else if (c == 1) v = q_now;
                    else if (c == 2) v = (float)data;
                    else if (c == 3) v = (float)wl;
                    else if (c == 4) v = (float)(rem > 0 ? rem : 0.0);
                    else { const int i = c - 5; v = s_ratio[i < dec ? i : i + 1]; }
                    s_obs[dec * w + c] = v;
                }
                n_pending += 1;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    STAMP(9);
    // ---- decay, clock, outputs (MRRL.PY:364-375,444-459) --------------------------------------------
    if constexpr (!RESET) {
        if (agent) {
            double l = r_load[sl] - cap_me * window;
            if (!(l > 0)) l = 0.0;
            r_load[sl] = l;
            const int q = r_queue[sl] - 1;
            r_queue[sl] = q > 0 ? q : 0;
        }
    }
    const uint64_t act_bits = group_bits<LPE>(__ballot(agent && r_rank[sl] >= 0), g);
    if (out.n_draws) {
        int d = draws_lane;
        for (int m = LPE / 2; m > 0; m >>= 1) d += __shfl_xor(d, m, LPE);
        if (valid && sl == 0) out.n_draws[env] = d + draws_env;
    }
    if (valid && sl == 0) {
        if constexpr (!RESET) {
            r_scal[S_TIME] = t_now + window;               // delta_t > 0 always: dt_min > 0
            r_scal[S_ELAPSED] = r_scal[S_ELAPSED] + window;
            r_int[I_STEP] = step_no;
            if (out.reward) out.reward[env] = reward;
            if (out.emitted) out.emitted[env] = n_pending > 0;
            if (out.reward_seen) out.reward_seen[env] = n_pending > 0 ? reward : 0.0;
            if (out.done) out.done[env] = step_no >= p.horizon;
        }
        r_int[I_PENDING] = n_pending;
        out.active[env] = act_bits;
    }
    if (agent && out.rank) out.rank[env * n + sl] = r_rank[sl];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    STAMP(10);
    // ---- stage out: records and observation tile, 16-byte lane-linear stores -------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(wbase);
        uint4* dst = reinterpret_cast<uint4*>(state + env0 * p.rec_bytes);
        for (int i = lane; i < n_valid * p.rec_bytes / 16; i += 64) dst[i] = src[i];
        const int64_t fbase = env0 * p.obs_floats;
        const int nfl = n_valid * p.obs_floats;
        float* gobs = out.obs + fbase;
        if ((fbase & 3) == 0) {
            const float4* s4 = reinterpret_cast<const float4*>(s_obs_w);
            float4* d4 = reinterpret_cast<float4*>(gobs);
            for (int i = lane; i < (nfl >> 2); i += 64) d4[i] = s4[i];
            for (int i = (nfl & ~3) + lane; i < nfl; i += 64) gobs[i] = s_obs_w[i];
        } else {
            for (int i = lane; i < nfl; i += 64) gobs[i] = s_obs_w[i];
        }
    }
    STAMP(11);
}

// ------------------------------------------------------------------------------------------------
// episode metrics (MRRL.PY:470-504); p99 from the latency log when present
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void metrics_kernel(const EnvParams p, const unsigned char* __restrict__ state,
                                                      const EnvConst* __restrict__ cc,
                                                      const double* __restrict__ lat_log, double* __restrict__ scratch,
                                                      double* __restrict__ out6)
{
    // one workgroup per environment: the p99 needs a sort of that environment's latencies
    const int64_t env = blockIdx.x;
    const int n = p.n;
    const unsigned char* rec = state + env * p.rec_bytes;
    const double* scal = reinterpret_cast<const double*>(rec + rec_off_scal(n));
    const int32_t* ri = reinterpret_cast<const int32_t*>(rec + rec_off_int(n));
    const int tasks = ri[I_TASKS];
    const int logged = lat_log ? ri[I_LOGGED] : 0;
    extern __shared__ __align__(16) unsigned char smem[];
    double* v = reinterpret_cast<double*>(smem);   // next pow2 >= logged doubles
    double p99 = __builtin_nan("");
    if (tasks == 0) p99 = 0.0;
    else if (lat_log && logged == tasks) {
        int m = 1;
        while (m < logged) m <<= 1;
        double* buf = (m * 8 <= 128 * 1024) ? v : scratch + env * p.lat_cap * 2;   // large logs sort in HBM scratch
        for (int i = threadIdx.x; i < m; i += blockDim.x)
            buf[i] = i < logged ? lat_log[env * p.lat_cap + i] * 1000 : __builtin_inf();
        __syncthreads();
        for (int size = 2; size <= m; size <<= 1) {
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = threadIdx.x; i < (m >> 1); i += blockDim.x) {
                    const int lo = 2 * i - (i & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = ((lo & size) == 0);
                    const double a = buf[lo], b = buf[hi];
                    if ((a > b) == up) { buf[lo] = b; buf[hi] = a; }
                }
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) {
            const double vi = (double)(logged - 1) * (99.0 / 100.0);   // numpy 'linear' method
            const double fl = floor(vi);
            int lo = (int)fl, hi = lo + 1;
            if (vi >= (double)(logged - 1)) { lo = logged - 1; hi = logged - 1; }
            if (hi > logged - 1) hi = logged - 1;
            const double a = buf[lo], b = buf[hi], t = vi - fl, d = b - a;
            p99 = (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;
        }
    }
    if (threadIdx.x == 0) {
        double* o = out6 + env * 6;
        const double cnt = (double)(tasks ? tasks : 1);
        o[0] = tasks ? scal[S_SUMLAT] * 1000 / (double)tasks : 0.0;
        o[1] = p99;
        o[2] = tasks ? scal[S_SUMEN] / (double)tasks : __builtin_nan("");
        o[3] = (double)ri[I_VIOL] / cnt;
        const double el = scal[S_ELAPSED];
        o[4] = (double)ri[I_DONE] / (el > 1e-6 ? el : 1e-6);
        // Jain index over per-node mean cpu usage; usage sample of a step = cap*dt/(cap+1e-9)
        // (MRRL.PY:379-381), reset contributes one zero sample (MRRL.PY:367-368,381).
        const double samples = (double)(ri[I_STEP] + 1);
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < n; ++i) {
            const double m_i = (cc->cap[i] * el / (cc->cap[i] + 1e-9)) / samples;
            s1 += m_i;
            s2 += m_i * m_i;
        }
        o[5] = (s1 * s1) / ((double)n * s2 + 1e-9);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// host side: handle and C ABI
// ------------------------------------------------------------------------------------------------
struct em_handle {
    em_config cfg;
    int64_t n_envs = 0;
    int device = 0;
    uint64_t seed = 0, env_offset = 0;
    uint32_t episode = 0;          // Philox episode index; first reset uses 1
    int64_t explore_counter = 0;
    int explore_off = 0;
    int64_t lat_cap = 0;
    int lanes = 0;                 // lanes per environment in use
    int lanes_forced = 0;
    unsigned char* d_state = nullptr;
    EnvConst* d_const = nullptr;
    double* d_lat_log = nullptr;
    double* d_sort_scratch = nullptr;
    int32_t* d_err = nullptr;
    bool was_reset = false;
};

namespace {

int pick_lanes(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }

EnvConst make_const(const em_config& c)
{
    EnvConst k;
    std::memset(&k, 0, sizeof k);
    k.rate_lo = c.arrival_rate_lo; k.rate_span = c.arrival_rate_hi - c.arrival_rate_lo;
    k.rate_mid = 0.5 * (c.arrival_rate_lo + c.arrival_rate_hi);
    k.cycle = c.arrival_cycle_seconds > 1e-6 ? c.arrival_cycle_seconds : 1e-6;
    k.inv_cycle = 1.0 / k.cycle;
    k.wl_lo = c.workload_lo; k.wl_span = c.workload_hi - c.workload_lo;
    k.data_lo = c.data_lo; k.data_span = c.data_hi - c.data_lo;
    k.ddl_lo = c.deadline_lo; k.ddl_span = c.deadline_hi - c.deadline_lo;
    k.bw_lo = c.bw_lo; k.bw_span = c.bw_hi - c.bw_lo;
    k.lat_lo = c.net_latency_lo; k.lat_span = c.net_latency_hi - c.net_latency_lo;
    k.w_lat = c.w_latency; k.w_en = c.w_energy; k.w_viol = c.w_violation;
    k.dt_min = c.min_time_step; k.dt_max = c.max_time_step; k.window0 = c.initial_observation_window;
    for (int i = 0; i < EM_MAX_NODES; ++i) {
        const double cap = i < c.n_nodes ? c.node_cpu_capacity[i] : 1.0;
        k.cap[i] = cap;
        k.inv_cap[i] = 1.0 / cap;
    }
    k.inv_k[0] = 0.0;
    for (int i = 1; i <= EM_MAX_NODES; ++i) k.inv_k[i] = 1.0 / (double)i;
    return k;
}

EnvParams make_params(const em_handle* h, int lanes)
{
    const em_config& c = h->cfg;
    EnvParams p;
    std::memset(&p, 0, sizeof p);
    p.n = c.n_nodes;
    p.horizon = c.episode_length;
    p.rate_mode = c.arrival_rate_lo == c.arrival_rate_hi ? 3 : c.arrival_mode;
    p.explore = h->explore_off ? 0 : 1;
    p.rec_bytes = rec_bytes(p.n);
    p.obs_floats = p.n * (p.n + 4);
    const int epw = 64 / lanes;
    int wb = epw * p.rec_bytes;
    wb += (epw * p.obs_floats * 4 + 15) & ~15;
    wb += (3 * p.n + 1) * 8;       // capacity, 1/capacity and 1/k tables
    wb += 2 * epw * p.n * 8;       // latencies, energies of the step
    wb += (epw * p.n * 4 + 7) & ~7;   // load ratios
    wb += epw * PRE_LANES * 20;    // pre-drawn environment-level random numbers
    p.wave_bytes = (wb + 15) & ~15;
    // _get_current_epsilon, MRRL.PY:249-256
    double eps = 0.0;
    if (!h->explore_off) {
        const int64_t span = c.decay_steps > 1 ? c.decay_steps : 1;
        double frac = (double)h->explore_counter / (double)span;
        if (frac > 1.0) frac = 1.0;
        eps = c.initial_epsilon - (c.initial_epsilon - c.final_epsilon) * frac;
    }
    p.eps = eps;
    p.seed_lo = (uint32_t)h->seed; p.seed_hi = (uint32_t)(h->seed >> 32);
    p.episode = h->episode;
    p.env_offset = (uint32_t)h->env_offset;
    p.n_envs = h->n_envs;
    p.lat_cap = h->lat_cap;
    return p;
}

template <bool TAPE, bool RESET>
int launch_env(em_handle* h, const int32_t* placement, const float* resource, const em_tape* tape,
               const em_step_out* out, hipStream_t s)
{
    const int lanes = h->lanes;
    const EnvParams p = make_params(h, lanes);
    const int epw = 64 / lanes;
    const int64_t waves = (h->n_envs + epw - 1) / epw;
    const int64_t blocks = (waves + 3) / 4;
    const size_t lds = (size_t)p.wave_bytes * 4;
    if (lds > 160 * 1024) return em::fail(EM_ERR_INVALID, "n_nodes too large for the LDS tile (%zu bytes)", lds);
    em_tape tp;
    std::memset(&tp, 0, sizeof tp);
    if (tape) tp = *tape;
    const dim3 grid((unsigned)blocks), block(256);
#define EM_LAUNCH(L)                                                                                              \
    hipLaunchKernelGGL((env_kernel<L, TAPE, RESET>), grid, block, lds, s, p, h->d_state, h->d_const, placement,     \
                       resource, tp, *out, h->d_lat_log, h->d_err)
    if (lds > 64 * 1024) {
        hipError_t e = hipSuccess;
        if (lanes == 16) e = hipFuncSetAttribute((const void*)env_kernel<16, TAPE, RESET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else if (lanes == 32) e = hipFuncSetAttribute((const void*)env_kernel<32, TAPE, RESET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        else e = hipFuncSetAttribute((const void*)env_kernel<64, TAPE, RESET>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return em::hip_fail(e, "hipFuncSetAttribute");
    }
    if (lanes == 16) EM_LAUNCH(16);
    else if (lanes == 32) EM_LAUNCH(32);
    else EM_LAUNCH(64);
#undef EM_LAUNCH
    return em::hip_check(hipGetLastError(), "env_kernel launch");
}

int check_out(const em_handle* h, const em_step_out* out, bool need_step_outputs)
{
    if (!out || !out->obs || !out->active) return em::fail(EM_ERR_INVALID, "obs and active outputs are required");
    if (((uintptr_t)out->obs & 15) != 0) return em::fail(EM_ERR_INVALID, "obs must be 16-byte aligned");
    (void)h; (void)need_step_outputs;
    return EM_OK;
}

int check_tape(const em_handle* h, const em_tape* t, bool reset)
{
    if (!t) return EM_OK;
    if (t->arrival_slots < 1 || !t->n_arrivals || !t->task_workload || !t->task_data || !t->task_deadline_offset ||
        !t->task_decider)
        return em::fail(EM_ERR_INVALID, "tape: arrival arrays missing");
    if (!reset && !t->dt_raw) return em::fail(EM_ERR_INVALID, "tape: dt_raw missing");
    if (h->cfg.arrival_mode == 2 && h->cfg.arrival_rate_lo != h->cfg.arrival_rate_hi && !t->rate_u)
        return em::fail(EM_ERR_INVALID, "tape: rate_u missing for random arrival mode");
    if (!reset && (!t->bw_u || !t->net_lat)) return em::fail(EM_ERR_INVALID, "tape: bw_u/net_lat missing");
    if (!reset && !h->explore_off && (!t->explore_u || !t->explore_tgt || !t->explore_res))
        return em::fail(EM_ERR_INVALID, "tape: exploration arrays missing while exploration is enabled");
    return EM_OK;
}

}  // namespace

extern "C" {

int em_abi_version(void) { return EM_ABI_VERSION; }

int em_create(const em_config* cfg, int64_t n_envs, int device_id, uint64_t seed, uint64_t env_offset,
              int64_t latency_log_capacity, em_handle** out)
{
    if (!cfg || !out) return em::fail(EM_ERR_INVALID, "em_create: null argument");
    if (cfg->n_nodes < 1 || cfg->n_nodes > EM_MAX_NODES) return em::fail(EM_ERR_INVALID, "n_nodes must be 1..%d", EM_MAX_NODES);
    if (n_envs < 1 || n_envs > (1ll << 31)) return em::fail(EM_ERR_INVALID, "n_envs out of range");
    if (cfg->episode_length < 1) return em::fail(EM_ERR_INVALID, "episode_length must be positive");
    if (!(cfg->min_time_step > 0) || !(cfg->max_time_step >= cfg->min_time_step))
        return em::fail(EM_ERR_INVALID, "need 0 < min_time_step <= max_time_step");
    if (latency_log_capacity < 0) return em::fail(EM_ERR_INVALID, "latency_log_capacity must be >= 0");
    for (int i = 0; i < cfg->n_nodes; ++i)
        if (!(cfg->node_cpu_capacity[i] > 0)) return em::fail(EM_ERR_INVALID, "node_cpu_capacity[%d] must be positive", i);
    EM_HIP(hipSetDevice(device_id));
    em_handle* h = new (std::nothrow) em_handle();
    if (!h) return em::fail(EM_ERR_INVALID, "out of host memory");
    h->cfg = *cfg;
    h->n_envs = n_envs;
    h->device = device_id;
    h->seed = seed;
    h->env_offset = env_offset;
    h->explore_off = cfg->disable_env_exploration;
    h->lat_cap = latency_log_capacity;
    h->lanes = pick_lanes(cfg->n_nodes);
    const size_t sbytes = (size_t)n_envs * rec_bytes(cfg->n_nodes);
    hipError_t e = hipMalloc(&h->d_state, sbytes);
    if (e == hipSuccess) e = hipMemset(h->d_state, 0, sbytes);
    if (e == hipSuccess) e = hipMalloc(&h->d_const, sizeof(EnvConst));
    if (e == hipSuccess) {
        const EnvConst k = make_const(*cfg);
        e = hipMemcpy(h->d_const, &k, sizeof k, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_err, sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(h->d_err, 0, sizeof(int32_t));
    if (e == hipSuccess && h->lat_cap > 0) {
        e = hipMalloc(&h->d_lat_log, sizeof(double) * (size_t)n_envs * (size_t)h->lat_cap);
        if (e == hipSuccess && h->lat_cap * 8 > 64 * 1024)
            e = hipMalloc(&h->d_sort_scratch, sizeof(double) * 2 * (size_t)n_envs * (size_t)h->lat_cap);
    }
    if (e != hipSuccess) {
        em_destroy(h);
        return em::hip_fail(e, "em_create allocation");
    }
    *out = h;
    return EM_OK;
}

int em_destroy(em_handle* h)
{
    if (!h) return EM_OK;
    (void)hipSetDevice(h->device);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_const) (void)hipFree(h->d_const);
    if (h->d_lat_log) (void)hipFree(h->d_lat_log);
    if (h->d_sort_scratch) (void)hipFree(h->d_sort_scratch);
    if (h->d_err) (void)hipFree(h->d_err);
    delete h;
    return EM_OK;
}

int em_reset(em_handle* h, const em_tape* tape, int has_seed, uint64_t seed, const em_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "em_reset: null handle");
    int rc = check_out(h, out, false);
    if (rc) return rc;
    rc = check_tape(h, tape, true);
    if (rc) return rc;
    if (has_seed) { h->seed = seed; h->episode = 0; }
    h->episode += 1;
    h->was_reset = true;
    hipStream_t s = (hipStream_t)stream;
    return tape ? launch_env<true, true>(h, nullptr, nullptr, tape, out, s)
                : launch_env<false, true>(h, nullptr, nullptr, nullptr, out, s);
}

int em_step(em_handle* h, const int32_t* placement, const float* resource, const em_tape* tape,
            const em_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "em_step: null handle");
    if (!h->was_reset) return em::fail(EM_ERR_STATE, "em_step before em_reset");
    if (!placement || !resource) return em::fail(EM_ERR_INVALID, "em_step: placement/resource are required");
    int rc = check_out(h, out, true);
    if (rc) return rc;
    rc = check_tape(h, tape, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = tape ? launch_env<true, false>(h, placement, resource, tape, out, s)
              : launch_env<false, false>(h, placement, resource, nullptr, out, s);
    if (rc) return rc;
    if (!h->explore_off) h->explore_counter += 1;   // MRRL.PY:467-468
    return EM_OK;
}

int em_episode_metrics(em_handle* h, double* out6, void* stream)
{
    if (!h || !out6) return em::fail(EM_ERR_INVALID, "em_episode_metrics: null argument");
    const EnvParams p = make_params(h, h->lanes);
    int64_t m = 1;
    while (m < h->lat_cap) m <<= 1;
    size_t lds = h->d_lat_log ? (size_t)m * 8 : 16;
    if (lds > 128 * 1024) lds = 16;
    if (lds > 64 * 1024)
        EM_HIP(hipFuncSetAttribute((const void*)metrics_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(metrics_kernel, dim3((unsigned)h->n_envs), dim3(256), lds, (hipStream_t)stream, p, h->d_state,
                       h->d_const, h->d_lat_log, h->d_sort_scratch, out6);
    return em::hip_check(hipGetLastError(), "metrics_kernel launch");
}

int em_set_exploration(em_handle* h, int disable_env_exploration, int64_t global_step_counter)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    if (global_step_counter < 0) return em::fail(EM_ERR_INVALID, "counter must be >= 0");
    h->explore_off = disable_env_exploration ? 1 : 0;
    h->explore_counter = global_step_counter;
    return EM_OK;
}

int64_t em_get_exploration_counter(const em_handle* h) { return h ? h->explore_counter : -1; }
int64_t em_record_bytes(const em_handle* h) { return h ? rec_bytes(h->cfg.n_nodes) : -1; }
int64_t em_n_envs(const em_handle* h) { return h ? h->n_envs : -1; }

int em_set_lanes_per_env(em_handle* h, int lanes)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    if (lanes == 0) { h->lanes = pick_lanes(h->cfg.n_nodes); return EM_OK; }
    if ((lanes != 16 && lanes != 32 && lanes != 64) || lanes < h->cfg.n_nodes)
        return em::fail(EM_ERR_INVALID, "lanes per env must be 16, 32 or 64 and >= n_nodes");
    h->lanes = lanes;
    return EM_OK;
}

int em_get_state(em_handle* h, double* load, double* queue, double* sim_time, double* elapsed, int32_t* step,
                 int64_t* counts, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    const int n = h->cfg.n_nodes, rb = rec_bytes(n);
    std::vector<unsigned char> host((size_t)h->n_envs * rb);
    EM_HIP(hipStreamSynchronize((hipStream_t)stream));
    EM_HIP(hipMemcpy(host.data(), h->d_state, host.size(), hipMemcpyDeviceToHost));
    int32_t err = 0;
    EM_HIP(hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (err) return em::fail(EM_ERR_INVALID, "device flagged malformed tape/action input (flags %d)", err);
    for (int64_t b = 0; b < h->n_envs; ++b) {
        const unsigned char* rec = host.data() + b * rb;
        const double* ld = reinterpret_cast<const double*>(rec);
        const double* sc = reinterpret_cast<const double*>(rec + rec_off_scal(n));
        const int32_t* q = reinterpret_cast<const int32_t*>(rec + rec_off_queue(n));
        const int32_t* ri = reinterpret_cast<const int32_t*>(rec + rec_off_int(n));
        for (int i = 0; i < n; ++i) {
            if (load) load[b * n + i] = ld[i];
            if (queue) queue[b * n + i] = (double)q[i];
        }
        if (sim_time) sim_time[b] = sc[S_TIME];
        if (elapsed) elapsed[b] = sc[S_ELAPSED];
        if (step) step[b] = ri[I_STEP];
        if (counts) { counts[3 * b] = ri[I_TASKS]; counts[3 * b + 1] = ri[I_VIOL]; counts[3 * b + 2] = ri[I_DONE]; }
    }
    return EM_OK;
}

}  // extern "C"
