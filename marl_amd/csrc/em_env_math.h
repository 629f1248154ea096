// em_env_math.h - arithmetic shared by the environment kernels (edge_env.hip, edge_env_scaled.hip): per-handle
// constants, exact division by constants, the Philox4x32-10 stream and the deterministic float64 log/exp/sin that
// oracle/philox_ref.py and oracle/edge_oracle.c mirror bit for bit.  Include from a translation unit compiled with
// -ffp-contract=off; everything lives in an anonymous namespace (one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "edgemarl.h"

#pragma clang fp contract(off)

namespace {

// Per-handle constants, resident in HBM (read with scalar loads where they are used, which keeps them
// out of the kernel's long-lived SGPRs).  inv_* are correctly rounded reciprocals computed on the host:
// a quotient by one of these constants is formed as q0 = x*inv; q = fma(fma(-q0, d, x), inv, q0), the
// final steps of the hardware's own float64 division sequence (v_div_fmas after the residual), which is
// correctly rounded for the value ranges of the simulator (tests/test_env_gpu.py::test_const_division).
struct EnvConst {
    double rate_lo, rate_span, rate_mid, cycle, inv_cycle;
    double wl_lo, wl_span, data_lo, data_span, ddl_lo, ddl_span, bw_lo, bw_span, lat_lo, lat_span;
    double w_lat, w_en, w_viol, dt_min, dt_max, window0;
    double eps_init, eps_final, eps_span;    // exploration schedule (MRRL.PY:249-256)
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

// np.percentile(x, 99) ('linear' method) from the two order statistics it interpolates: a = sorted[lo], b = sorted[hi]
__device__ inline void p99_indices(int logged, int& lo, int& hi, double& t)
{
    const double vi = (double)(logged - 1) * (99.0 / 100.0);
    const double fl = floor(vi);
    lo = (int)fl; hi = lo + 1;
    if (vi >= (double)(logged - 1)) { lo = logged - 1; hi = logged - 1; }
    if (hi > logged - 1) hi = logged - 1;
    t = vi - fl;
}
__device__ inline double p99_lerp(double a, double b, double t)
{
    const double d = b - a;
    return (t >= 0.5) ? b - d * (1.0 - t) : a + d * t;
}

// p99 (in ms) of one environment's latency log (seconds) by SELECTION: the log sits in one wavefront's registers (PL values
// per lane, logged <= 64 PL) and maxima are extracted one at a time (lane-local maximum, wavefront maximum by shuffles, the
// lowest owning lane drops one copy) down to the two order statistics np.percentile interpolates, about 1 % of the log
// from the top.  All 64 lanes call it; same values as sorting.  NaN if the log does not fit (logged > 64 PL).
template <int PL>
__device__ inline double wave_p99(const double* __restrict__ log, int logged, int lane)
{
    int lo, hi;
    double t;
    p99_indices(logged, lo, hi, t);
    const int k_lo = logged - 1 - lo, k_hi = logged - 1 - hi;      // ranks from the top (0 = maximum), k_hi <= k_lo
    if (logged > 64 * PL) return __builtin_nan("");
    double v[PL];
#pragma unroll
    for (int j = 0; j < PL; ++j) {
        const int i = j * 64 + lane;
        v[j] = i < logged ? log[i] * 1000 : -__builtin_inf();
    }
    double a = 0.0, b = 0.0;
    for (int r = 0; r <= k_lo; ++r) {
        double m = v[0];
#pragma unroll
        for (int j = 1; j < PL; ++j) m = v[j] > m ? v[j] : m;
        double w = m;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const double o = __shfl_xor(w, d); w = o > w ? o : w; }
        if (r == k_hi) b = w;
        if (r == k_lo) a = w;
        const unsigned long long own = __ballot(m == w);
        if (lane == __ffsll(own) - 1) {                            // drop ONE copy of the maximum
            bool gone = false;
#pragma unroll
            for (int j = 0; j < PL; ++j)
                if (!gone && v[j] == w) { v[j] = -__builtin_inf(); gone = true; }
        }
    }
    return p99_lerp(a, b, t);
}

// The same two order statistics for a log that does not fit the registers: distinct values are visited from the top, each
// visit one pass over the log in memory (maximum below the previous value, and how many copies of it there are).  One pass
// per distinct value among the top 1 %: slow, and only taken by an environment that logged more than 64 PL tasks.
__device__ inline double wave_p99_stream(const double* __restrict__ log, int logged, int lane)
{
    int lo, hi;
    double t;
    p99_indices(logged, lo, hi, t);
    const int k_lo = logged - 1 - lo, k_hi = logged - 1 - hi;
    double a = 0.0, b = 0.0, thr = __builtin_inf();
    int pos = 0;
    while (pos <= k_lo) {
        double m = -__builtin_inf();
        for (int i = lane; i < logged; i += 64) { const double x = log[i] * 1000; if (x < thr && x > m) m = x; }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const double o = __shfl_xor(m, d); m = o > m ? o : m; }
        int c = 0;
        for (int i = lane; i < logged; i += 64) c += (log[i] * 1000 == m) ? 1 : 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
        if (c <= 0) break;                                         // cannot happen for finite latencies; no endless loop on NaN
        if (k_hi >= pos && k_hi < pos + c) b = m;
        if (k_lo >= pos && k_lo < pos + c) a = m;
        pos += c;
        thr = m;
    }
    return p99_lerp(a, b, t);
}

template <int PL>
__device__ inline double wave_p99_any(const double* __restrict__ log, int logged, int lane)
{
    return logged <= 64 * PL ? wave_p99<PL>(log, logged, lane) : wave_p99_stream(log, logged, lane);   // wave-uniform
}

inline EnvConst make_const(const em_config& c)
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
    k.eps_init = c.initial_epsilon; k.eps_final = c.final_epsilon;
    k.eps_span = (double)(c.decay_steps > 1 ? c.decay_steps : 1);
    for (int i = 0; i < EM_MAX_NODES; ++i) {
        const double cap = i < c.n_nodes ? c.node_cpu_capacity[i] : 1.0;
        k.cap[i] = cap;
        k.inv_cap[i] = 1.0 / cap;
    }
    k.inv_k[0] = 0.0;
    for (int i = 1; i <= EM_MAX_NODES; ++i) k.inv_k[i] = 1.0 / (double)i;
    return k;
}

}  // namespace
