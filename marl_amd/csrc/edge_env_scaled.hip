// edge_env_scaled.hip - the EdgeSimGym step for the scaled topology: D decider agents
// over S <= 64 edge servers, one WAVEFRONT per environment, plus its C ABI (ems_*).
//
// The step is the reference's (MRRL.PY:188-521) with "agent == server" split into decider d and its home server
// d mod S (include/edgemarl.h states the generalisation; D == S is the reference bit for bit).  Mapping:
//   * servers on lanes: load/queue of server s live on lane s (registers + an LDS copy for indexed access);
//   * the pending list (deciders holding a task, arrival order) on lanes: slot k on lane k, at most 64;
//   * the arrivals of the step on lanes: arrival j on lane j (more than 64 take a sequential tail).
// The order-dependent part of the placement loop (MRRL.PY:416,437: agent k sees the load agent k-1 added) is a
// loop over acting slots in which one lane touches the LDS copy of load[target]; everything else is lane parallel.
// Outputs are compact (pending list + observation rows), the state record is read and written once per step.
// Two things keep the wavefront's instruction count down (the step is VALU-issue bound, not HBM bound, when every
// lane of a wavefront repeats per-environment scalar work): (1) the per-environment scalar chain of
// _generate_tasks_and_obs (rate -> window -> Poisson count) depends on the clock only, never on the actions, so
// scaled_chain_kernel evaluates it one environment per LANE for CHAIN_ROWS steps ahead; (2) all Philox blocks a step
// needs (arrivals, network draws, exploration) are evaluated in ONE pass with a lane-dependent counter.
// All environment arithmetic is float64 in the reference's evaluation order (-ffp-contract=off).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "edgemarl.h"
#include "em_common.h"
#include "em_env_math.h"

#pragma clang fp contract(off)

namespace {

#ifndef EMS_MIN_WAVES
#define EMS_MIN_WAVES 5      // wavefronts per SIMD the register allocation aims for (one environment per wavefront)
#endif
#ifndef EMS_MIN_WAVES4
#define EMS_MIN_WAVES4 4     // same, four environments per wavefront
#endif
enum { PCAP = EMS_PENDING_CAP, WAVES_PER_BLOCK = 4, PRE_POIS = 6, CHAIN_ROWS = 16 };
enum { SI_STEP = 0, SI_TASKS, SI_VIOL, SI_DONE, SI_PENDING, SI_LOGGED };
enum { SS_TIME = 0, SS_ELAPSED, SS_SUMLAT, SS_SUMEN };

// state record of one environment (HBM, a whole number of 128-byte lines):
//   f64 load[S] | i32 queue[S] | f64 sim_time, elapsed, sum_lat, sum_energy | i32 step, tasks, viol, completed, pending, logged, 0, 0
//   | pending list, 64 entries of 32 bytes in arrival order: {f64 workload, f64 data, f64 deadline, i32 decider, i32 0}
// A step reads the first k entries (k = 2.2 on average at 64 x 256): as entries they share one or two lines with the
// scalars in front of them.  (Until round 3 the list was four slot-major arrays - a line of each per step for 63 useful
// bytes - and records of 2,624 bytes started mid-line every other environment: the HBM counters showed 1.96 x the
// algorithmic bytes on the read side.)
struct PendEntry { double wl, data, ddl; int32_t dec, pad; };
static_assert(sizeof(PendEntry) == 32, "pending entries are 32 bytes");
__host__ __device__ inline int srec_off_queue(int S) { return 8 * S; }
__host__ __device__ inline int srec_off_scal(int S) { return (12 * S + 31) & ~31; }
__host__ __device__ inline int srec_off_int(int S) { return srec_off_scal(S) + 32; }
__host__ __device__ inline int srec_off_pend(int S) { return srec_off_int(S) + 32; }
__host__ __device__ inline int srec_bytes(int S) { return (srec_off_pend(S) + 32 * PCAP + 127) & ~127; }

struct ScaledParams {
    int32_t S, D, horizon, rate_mode, explore, rec_bytes, by_slot, step_no;
    double eps;
    uint32_t seed_lo, seed_hi, episode, env_offset;
    int64_t n_envs, lat_cap;
};

__device__ inline uint64_t lanes_below(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }

// The per-environment scalar chain of _generate_tasks_and_obs: arrival rate -> time window -> Poisson arrival
// count (MRRL.PY:220-247,350-351), identical on every lane of the environment's wavefront.
template <bool RESET>
__device__ inline void arrival_chain(const ScaledParams& p, const EnvConst* __restrict__ cc, uint32_t eid, uint32_t step_no,
                                     double t0, double& window, int& n_new)
{
    const uint4 b = philox(eid, p.episode, step_no, (P_DT << 24), p.seed_lo, p.seed_hi);
    const double dt_u = u53(b.x, b.y);
    const double rate_u = cc->rate_lo + cc->rate_span * u53(b.z, b.w);
    double pu[PRE_POIS];
#pragma unroll
    for (int i = 0; i < PRE_POIS / 2; ++i) {
        const uint4 q = philox(eid, p.episode, step_no, (P_POIS << 24) | (uint32_t)i, p.seed_lo, p.seed_hi);
        pu[2 * i] = u53(q.x, q.y);
        pu[2 * i + 1] = u53(q.z, q.w);
    }
    double rate;
    if (p.rate_mode == 3) rate = cc->rate_lo;                         // degenerate range (MRRL.PY:222-223)
    else if (p.rate_mode == 1) {
        const double cyc = cc->cycle, icyc = cc->inv_cycle;           // clamped to >= 1e-6 on the host (MRRL.PY:226)
        const double phase = qdiv(exact_fmod_pos(t0, cyc, icyc), cyc, icyc);
        rate = cc->rate_lo + (0.5 + 0.5 * det_sin2pi(phase)) * cc->rate_span;
    } else if (p.rate_mode == 2) rate = rate_u;
    else rate = cc->rate_mid;
    const double rate_c = rate > 1e-6 ? rate : 1e-6;
    if constexpr (RESET) window = cc->window0;
    else {
        const double raw = (1.0 / rate_c) * (-det_log(1.0 - dt_u));
        const double dt_min = cc->dt_min, dt_max = cc->dt_max;
        window = raw < dt_min ? dt_min : (raw > dt_max ? dt_max : raw);   // MRRL.PY:247
    }
    const double limit = det_exp(-(rate_c * window));
    double prod = 1.0;
    bool going = true;
    n_new = 0;
#pragma unroll
    for (int i = 0; i < PRE_POIS; ++i) {     // Knuth's multiplication method on the pre-drawn uniforms
        prod = prod * pu[i];
        going = going && prod > limit;
        n_new += going ? 1 : 0;
    }
    if (going) {
        uint4 q = make_uint4(0, 0, 0, 0);
        for (uint32_t i = PRE_POIS;; ++i) {  // ends at n_new == 4096 at the latest
            if ((i & 1u) == 0u) q = philox(eid, p.episode, step_no, (P_POIS << 24) | (i >> 1), p.seed_lo, p.seed_hi);
            const double u = (i & 1u) ? u53(q.z, q.w) : u53(q.x, q.y);
            prod = prod * u;
            if (prod > limit && n_new < 4096) ++n_new; else break;
        }
    }
}

__device__ inline void draw_task(const ScaledParams& p, const EnvConst* __restrict__ cc, uint32_t eid, uint32_t step_no, uint32_t j,
                                 double& wl, double& data, double& offs, int& dec)
{
    const uint4 a = philox(eid, p.episode, step_no, (P_TASK_A << 24) | j, p.seed_lo, p.seed_hi);
    const uint4 b = philox(eid, p.episode, step_no, (P_TASK_B << 24) | j, p.seed_lo, p.seed_hi);
    wl = cc->wl_lo + cc->wl_span * u53(a.x, a.y);
    data = cc->data_lo + cc->data_span * u53(a.z, a.w);
    offs = cc->ddl_lo + cc->ddl_span * u53(b.x, b.y);
    dec = (int)__umulhi(b.z, (uint32_t)p.D);
}

__device__ inline int rl_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline float rl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ inline double rl_d(double v, int l)
{
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// The scalar chain for `rows` consecutive steps, one environment per lane.  Row r holds step first_step + r; step 0 is
// the reset (fixed window, clock does not advance: MRRL.PY:209,345-348).  `clock` carries each environment's
// sim_time between launches: t(k+1) = t(k) + window(k) is the very sum the step kernel stores (MRRL.PY:444-447).
__global__ __launch_bounds__(256) void scaled_chain_kernel(const ScaledParams p, const EnvConst* __restrict__ cc,
                                                           double* __restrict__ clock, double* __restrict__ win_out,
                                                           int32_t* __restrict__ n_out, int first_step, int rows)
{
    const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (env >= p.n_envs) return;
    const uint32_t eid = p.env_offset + (uint32_t)env;
    double t = first_step == 0 ? 0.0 : clock[env];
    for (int r = 0; r < rows; ++r) {
        const uint32_t step_no = (uint32_t)(first_step + r);
        double window;
        int n_new;
        if (step_no == 0u) arrival_chain<true>(p, cc, eid, step_no, t, window, n_new);
        else {
            arrival_chain<false>(p, cc, eid, step_no, t, window, n_new);
            t = t + window;
        }
        win_out[(int64_t)r * p.n_envs + env] = window;
        n_out[(int64_t)r * p.n_envs + env] = n_new;
    }
    clock[env] = t;
}

// The step of ONE environment on one wavefront (servers, pending slots and arrivals on lanes).  Used by
// scaled_env_kernel (one wavefront per environment) and as the general path of scaled_env4_kernel for environments whose
// step has more than 16 pending slots or arrivals.  window / n_new: this step's row of the scalar chain.
// SC / DC: server and decider counts as compile-time constants (0: taken from the launch parameters); instantiated for
// BASELINE.json configs[4] (64 servers, 256 deciders) so that record offsets, row widths and loop bounds fold.
template <bool RESET, int SC = 0, int DC = 0>
__device__ __forceinline__ void step_one_env(const ScaledParams& p, unsigned char* __restrict__ state, const EnvConst* __restrict__ cc,
                                             const int32_t* __restrict__ placement, const float* __restrict__ resource,
                                             const double window, const int n_new, const ems_step_out& out,
                                             double* __restrict__ lat_log, int32_t* __restrict__ err_flag, const int64_t env,
                                             const int lane, double* s_load, double* s_lat, double* s_en, int32_t* s_queue,
                                             float* s_ratio, int32_t* s_pdec)
{
    const int S = SC ? SC : p.S, D = DC ? DC : p.D, w = S + 4;
    const int RB = SC ? srec_bytes(SC) : p.rec_bytes;
    const uint32_t eid = p.env_offset + (uint32_t)env;
    const uint32_t step_no = RESET ? 0u : (uint32_t)p.step_no;
    unsigned char* rec = state + env * (int64_t)RB;
    double* g_load = reinterpret_cast<double*>(rec);
    int32_t* g_queue = reinterpret_cast<int32_t*>(rec + srec_off_queue(S));
    double* g_scal = reinterpret_cast<double*>(rec + srec_off_scal(S));
    int32_t* g_int = reinterpret_cast<int32_t*>(rec + srec_off_int(S));
    PendEntry* g_pend = reinterpret_cast<PendEntry*>(rec + srec_off_pend(S));
    const uint64_t below = lanes_below(lane);

    // ---- global loads first ------------------------------------------------------------------------------------
    const bool srv = lane < S;
    const double cap_me = srv ? cc->cap[lane] : 1.0, icap_me = srv ? cc->inv_cap[lane] : 1.0;
    double load_me = 0.0, t_now = 0.0, el_now = 0.0, acc_lat = 0.0, acc_en = 0.0;
    int q_me = 0, k_old = 0, tasks_before = 0, n_viol_before = 0, n_done_before = 0;
    if constexpr (!RESET) {
        if (srv) { load_me = g_load[lane]; q_me = g_queue[lane]; }
        t_now = g_scal[SS_TIME]; el_now = g_scal[SS_ELAPSED]; acc_lat = g_scal[SS_SUMLAT]; acc_en = g_scal[SS_SUMEN];
        tasks_before = g_int[SI_TASKS]; n_viol_before = g_int[SI_VIOL]; n_done_before = g_int[SI_DONE];
        k_old = g_int[SI_PENDING];
    }
    // pending entry of slot `lane` and its action
    int dec = 0, tgt_in = -1;
    float res_in = 1.0f;
    double wl = 1.0, data = 1.0, ddl = 0.0;
    const bool have = !RESET && lane < k_old;
    if constexpr (!RESET) {
        if (have) {
            const PendEntry pe = g_pend[lane];
            dec = pe.dec; wl = pe.wl; data = pe.data; ddl = pe.ddl;
            const int64_t ai = p.by_slot ? env * PCAP + lane : env * (int64_t)D + dec;
            tgt_in = placement[ai];
            res_in = resource[ai];
        }
    }

    // ---- every Philox block of the step in one pass per 8 arrivals / 16 pending slots ---------------------------
    // producer lanes 0-7: P_TASK_A of arrival 8q+i, 8-15: P_TASK_B, 16-31: P_NET of the decider in slot 16q+i,
    // 32-47: P_EXPL, 48-63: P_EXPL_RES; the consumers (arrival j on lane j, slot k on lane k) fetch by shuffle.  The
    // stream is counter addressed, so drawing a number the reference would not have drawn changes nothing.
    double a_wl = 0.0, a_data = 0.0, a_offs = 0.0, net_bw = 0.0, net_lat = 0.0, ex_u = 2.0, ex_res = 0.0;
    int a_dec = 0, ex_tgt = 0;
    {
        const int n_par = n_new < 64 ? n_new : 64;
        const int pa = (n_par + 7) >> 3, ps = (k_old + 15) >> 4;
        const int n_pass = pa > ps ? pa : ps;
        for (int q = 0; q < n_pass; ++q) {
            const int d_slot = __shfl(dec, (16 * q + (lane & 15)) & 63);
            uint32_t word;
            if (lane < 8) word = (P_TASK_A << 24) | (uint32_t)(8 * q + lane);
            else if (lane < 16) word = (P_TASK_B << 24) | (uint32_t)(8 * q + lane - 8);
            else word = ((lane < 32 ? P_NET : (lane < 48 ? P_EXPL : P_EXPL_RES)) << 24) | (uint32_t)d_slot;
            const uint4 b = philox(eid, p.episode, step_no, word, p.seed_lo, p.seed_hi);
            const double d0 = u53(b.x, b.y), d1 = u53(b.z, b.w);
            const int sa = lane & 7, ss = lane & 15;
            const double ta0 = __shfl(d0, sa), ta1 = __shfl(d1, sa), tb0 = __shfl(d0, 8 + sa);
            const uint32_t tbz = (uint32_t)__shfl((int)b.z, 8 + sa);
            const double nn0 = __shfl(d0, 16 + ss), nn1 = __shfl(d1, 16 + ss), ee0 = __shfl(d0, 32 + ss), rr0 = __shfl(d0, 48 + ss);
            const uint32_t eez = (uint32_t)__shfl((int)b.z, 32 + ss);
            if ((lane >> 3) == q) {
                a_wl = cc->wl_lo + cc->wl_span * ta0;
                a_data = cc->data_lo + cc->data_span * ta1;
                a_offs = cc->ddl_lo + cc->ddl_span * tb0;
                a_dec = (int)__umulhi(tbz, (uint32_t)D);
            }
            if ((lane >> 4) == q) {
                net_bw = cc->bw_lo + cc->bw_span * nn0;
                net_lat = cc->lat_lo + cc->lat_span * nn1;
                ex_u = ee0;
                ex_tgt = (int)__umulhi(eez, (uint32_t)S);
                ex_res = 0.1 + (1.0 - 0.1) * rr0;
            }
        }
    }

    double reward = 0.0;
    if constexpr (!RESET) {
        // ---- placement phase (MRRL.PY:397-438) ----------------------------------------------------------------
        s_load[lane] = load_me;
        s_queue[lane] = q_me;
        int tgt = tgt_in;
        const bool acting = have && tgt >= 0 && tgt < S;
        float res = acting ? res_in : 1.0f;
        const uint64_t act_bits = __ballot(acting);
        const int slot = __popcll(act_bits & below);
        const int k = __popcll(act_bits);
        if (acting) {
            if (p.explore && !(ex_u > p.eps)) {   // _maybe_apply_env_exploration, MRRL.PY:258-274
                tgt = ex_tgt;
                res = (float)ex_res;
            }
        }
        const int home = dec % S;
        double t_trans = 0.0;
        if (acting && tgt != home) {         // MRRL.PY:409-414
            const double bw = net_bw * 1e6 / 8;
            t_trans = (data / bw) + net_lat;
        }
        const int tl = acting ? tgt : lane;
        const double cap_t = __shfl(cap_me, tl), icap_t = __shfl(icap_me, tl);
        const double alloc = cap_t * (double)res;        // MRRL.PY:418 (float64 * float32)
        const double gcyc = qdiv(alloc, 1e9, 1e-9);
        const double t_exec = wl / gcyc;                  // MRRL.PY:419
        const double energy = wl * gcyc;                  // MRRL.PY:423
        const double inc = t_exec * gcyc;                 // MRRL.PY:437
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // ordered chain over acting slots: the queueing delay sees earlier placements (MRRL.PY:416,437)
        bool viol = false;
        double t_total = 0.0;
        for (int s = 0; s < k; ++s) {
            if (acting && slot == s) {
                const double l = s_load[tgt];
                const double t_queue = qdiv(l, cap_t, icap_t);
                t_total = t_trans + t_queue + t_exec;
                viol = (t_now + t_total) > ddl;          // MRRL.PY:431
                s_lat[s] = t_total;
                s_en[s] = energy;
                s_load[tgt] = l + inc;
                s_queue[tgt] += 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const int n_viol = __popcll(__ballot(viol));

        // reward (MRRL.PY:326-338) and episode accumulators (MRRL.PY:463-466)
        if (k > 0) {
            const double sum_lat = np_sum_small(s_lat, k), sum_en = np_sum_small(s_en, k);
            for (int i = 0; i < k; ++i) { acc_lat += s_lat[i]; acc_en += s_en[i]; }
            const double kd = (double)k, ik = cc->inv_k[k];
            const double cost = (cc->w_lat * qdiv(sum_lat, kd, ik)) + (cc->w_en * qdiv(sum_en, kd, ik)) +
                                (cc->w_viol * (double)n_viol);
            reward = -cost;
        }
        if (out.step_latency) out.step_latency[env * PCAP + lane] = lane < k ? s_lat[lane] : __builtin_nan("");
        if (lat_log && acting) {
            const int64_t at = (int64_t)tasks_before + slot;
            if (at < p.lat_cap) lat_log[env * p.lat_cap + at] = t_total;
        }
        tasks_before += k;
        n_viol_before += n_viol;
        n_done_before += k - n_viol;
        load_me = s_load[lane];
        q_me = s_queue[lane];
    } else {
        s_load[lane] = 0.0;
        s_queue[lane] = 0;
    }

    // ---- arrivals (MRRL.PY:340-383) ---------------------------------------------------------------------------
    const float ratio_me = srv ? (float)qdiv(load_me, cap_me, icap_me) : 0.0f;   // load/cap BEFORE this step's decay
    s_ratio[lane] = ratio_me;
    const bool has = lane < n_new;
    const double a_deadline = t_now + a_offs;              // MRRL.PY:217
    const int a_home = a_dec % S;
    // a later arrival of a decider that already got a task this step only bumps the queue (MRRL.PY:358-359)
    bool dup = false;
    const int npar = n_new < 64 ? n_new : 64;
    for (int i = 0; i + 1 < npar; ++i) {
        const int di = rl_i(a_dec, i);
        dup = dup || (lane > i && a_dec == di);
    }
    const bool is_new = has && !dup;
    const uint64_t new_bits = __ballot(is_new);
    uint64_t dup_bits = __ballot(has && dup);
    const int rank = __popcll(new_bits & below);
    int n_pending = __popcll(new_bits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // queue length the new decider observes (MRRL.PY:311): after the placements, plus the bumps that earlier
    // arrivals of this step made to its home server (several deciders share a home when D > S)
    int q_obs = is_new ? s_queue[a_home] : 0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    while (dup_bits != 0ull) {
        const int i = __ffsll((unsigned long long)dup_bits) - 1;
        dup_bits &= dup_bits - 1ull;
        const int hi = rl_i(a_home, i);
        if (is_new && lane > i && a_home == hi) q_obs += 1;
        if (lane == hi) q_me += 1;                          // lane hi < S owns server hi
    }
    if (is_new) s_pdec[rank] = a_dec;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // observation rows of the new pending deciders (_get_obs, MRRL.PY:307-324), compact, in arrival order
    const double a_rem = a_deadline - t_now;
    const float f_data = (float)a_data, f_wl = (float)a_wl, f_rem = (float)(a_rem > 0 ? a_rem : 0.0);
    {
        uint64_t nb = new_bits;
        int r = 0;
        while (nb != 0ull) {
            const int j = __ffsll((unsigned long long)nb) - 1;
            nb &= nb - 1ull;
            const int h = rl_i(a_home, j);
            const float c1 = (float)rl_i(q_obs, j), c2 = rl_f(f_data, j), c3 = rl_f(f_wl, j), c4 = rl_f(f_rem, j);
            float* row = out.pend_obs + (env * PCAP + r) * (int64_t)w;
            for (int c = lane; c < w; c += 64) {
                float v;
                if (c == 0) v = s_ratio[h];
                else if (c == 1) v = c1;
                else if (c == 2) v = c2;
                else if (c == 3) v = c3;
                else if (c == 4) v = c4;
                else { const int i = c - 5; v = s_ratio[i < h ? i : i + 1]; }
                row[c] = v;
            }
            ++r;
        }
    }
    if (is_new) {                                           // MRRL.PY:361: the decider's pending task
        g_pend[rank] = PendEntry{a_wl, a_data, a_deadline, a_dec, 0};
        out.pend_decider[env * PCAP + rank] = a_dec;
    }

    if (n_new > 64) {
        // rare tail: arrivals beyond one per lane, one at a time in arrival order.  s_queue holds the queue lengths
        // of the moment (placements + bumps so far); lane r < n_pending holds pending decider r for the busy test.
        s_queue[lane] = q_me;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int j = 64; j < n_new; ++j) {
            double twl, tdata, toffs;
            int tdec;
            draw_task(p, cc, eid, step_no, (uint32_t)j, twl, tdata, toffs, tdec);
            const int th = tdec % S;
            const bool busy = __ballot(lane < n_pending && lane < PCAP && s_pdec[lane] == tdec) != 0ull;
            const int q_now = s_queue[th];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (busy) {
                if (lane == 0) s_queue[th] = q_now + 1;
            } else if (n_pending >= PCAP) {
                if (lane == 0) atomicOr(err_flag, 16);      // more than 64 deciders hold a task: not representable
            } else {
                const double deadline = t_now + toffs, rem_d = deadline - t_now;
                float* row = out.pend_obs + (env * PCAP + n_pending) * (int64_t)w;
                for (int c = lane; c < w; c += 64) {
                    float v;
                    if (c == 0) v = s_ratio[th];
                    else if (c == 1) v = (float)q_now;
                    else if (c == 2) v = (float)tdata;
                    else if (c == 3) v = (float)twl;
                    else if (c == 4) v = (float)(rem_d > 0 ? rem_d : 0.0);
                    else { const int i = c - 5; v = s_ratio[i < th ? i : i + 1]; }
                    row[c] = v;
                }
                if (lane == 0) {
                    g_pend[n_pending] = PendEntry{twl, tdata, deadline, tdec, 0};
                    out.pend_decider[env * PCAP + n_pending] = tdec;
                    s_pdec[n_pending] = tdec;
                }
                n_pending += 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        q_me = s_queue[lane];
    }

    // ---- decay, clock, outputs (MRRL.PY:364-375,444-459) ---------------------------------------------------------
    if constexpr (!RESET) {
        double l = load_me - cap_me * window;
        if (!(l > 0)) l = 0.0;
        load_me = l;
        q_me = q_me - 1 > 0 ? q_me - 1 : 0;
    }
    if (srv) { g_load[lane] = load_me; g_queue[lane] = q_me; }
    if (lane == 0) {
        if constexpr (!RESET) {
            g_scal[SS_TIME] = t_now + window;
            g_scal[SS_ELAPSED] = el_now + window;
            g_scal[SS_SUMLAT] = acc_lat;
            g_scal[SS_SUMEN] = acc_en;
            g_int[SI_STEP] = (int32_t)step_no;
            g_int[SI_TASKS] = tasks_before;
            g_int[SI_VIOL] = n_viol_before;
            g_int[SI_DONE] = n_done_before;
            if (lat_log) g_int[SI_LOGGED] = (int32_t)((int64_t)tasks_before < p.lat_cap ? (int64_t)tasks_before : p.lat_cap);
            if (out.reward) out.reward[env] = reward;
            if (out.emitted) out.emitted[env] = n_pending > 0;
            if (out.reward_seen) out.reward_seen[env] = n_pending > 0 ? reward : 0.0;
            if (out.done) out.done[env] = (int32_t)step_no >= p.horizon;
        } else {
            g_scal[SS_TIME] = 0.0; g_scal[SS_ELAPSED] = 0.0; g_scal[SS_SUMLAT] = 0.0; g_scal[SS_SUMEN] = 0.0;
            g_int[SI_STEP] = 0; g_int[SI_TASKS] = 0; g_int[SI_VIOL] = 0; g_int[SI_DONE] = 0; g_int[SI_LOGGED] = 0;
        }
        g_int[SI_PENDING] = n_pending;
        out.n_pending[env] = n_pending;
    }
}

// One wavefront per environment; a workgroup is WAVES_PER_BLOCK independent wavefronts (no workgroup barrier).
// chain_win / chain_n: this step's row of the scalar chain (scaled_chain_kernel).
template <bool RESET, int SC = 0, int DC = 0>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, EMS_MIN_WAVES) void scaled_env_kernel(const ScaledParams p, unsigned char* __restrict__ state,
                                                                        const EnvConst* __restrict__ cc,
                                                                        const int32_t* __restrict__ placement,
                                                                        const float* __restrict__ resource,
                                                                        const double* __restrict__ chain_win,
                                                                        const int32_t* __restrict__ chain_n,
                                                                        const ems_step_out out, double* __restrict__ lat_log,
                                                                        int32_t* __restrict__ err_flag)
{
    __shared__ double s_load_all[WAVES_PER_BLOCK][64];
    __shared__ double s_lat_all[WAVES_PER_BLOCK][64];
    __shared__ double s_en_all[WAVES_PER_BLOCK][64];
    __shared__ int32_t s_queue_all[WAVES_PER_BLOCK][64];
    __shared__ float s_ratio_all[WAVES_PER_BLOCK][64];
    __shared__ int32_t s_pdec_all[WAVES_PER_BLOCK][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t env = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (env >= p.n_envs) return;                 // whole wavefronts leave; nothing below synchronises across wavefronts
    step_one_env<RESET, SC, DC>(p, state, cc, placement, resource, chain_win[env], chain_n[env], out, lat_log, err_flag, env, lane,
                        s_load_all[wave], s_lat_all[wave], s_en_all[wave], s_queue_all[wave], s_ratio_all[wave], s_pdec_all[wave]);
}

// FOUR environments per wavefront: the lane-sparse phases of the step (pending slots and arrivals: ~2 of them per
// environment) run on 16-lane groups, one group per environment, so their instructions are shared by four environments;
// the server phases (64 servers of one environment = one wavefront-wide operation) loop over the four environments.
// An environment whose step has more than 16 pending slots or more than 16 arrivals (rare: Poisson means are 1..3, 10 at
// reset) takes step_one_env on the whole wavefront afterwards.  Bit-identical to scaled_env_kernel.
constexpr int EPW4 = 4, GW4 = 16;
#ifdef EMS_STAMP      // cycle stamps of one wavefront of the step kernel (tools/stamp_scaled.py); compiled out of the library
__device__ unsigned long long ems_stamps[16];
#define ESTAMP(i) do { if (!RESET && blockIdx.x == 7 && threadIdx.x == 0) ems_stamps[i] = __builtin_readcyclecounter(); } while (0)
#else
#define ESTAMP(i) do { } while (0)
#endif

template <bool RESET, int SC = 0, int DC = 0>
__global__ __launch_bounds__(64 * WAVES_PER_BLOCK, EMS_MIN_WAVES4) void scaled_env4_kernel(const ScaledParams p, unsigned char* __restrict__ state,
                                                                         const EnvConst* __restrict__ cc,
                                                                         const int32_t* __restrict__ placement,
                                                                         const float* __restrict__ resource,
                                                                         const double* __restrict__ chain_win,
                                                                         const int32_t* __restrict__ chain_n,
                                                                         const ems_step_out out, double* __restrict__ lat_log,
                                                                         int32_t* __restrict__ err_flag)
{
    __shared__ double s_load_all[WAVES_PER_BLOCK][EPW4][64];
    __shared__ double s_lat_all[WAVES_PER_BLOCK][EPW4 * GW4];
    __shared__ double s_en_all[WAVES_PER_BLOCK][EPW4 * GW4];
    __shared__ int32_t s_queue_all[WAVES_PER_BLOCK][EPW4][64];
    __shared__ float s_ratio_all[WAVES_PER_BLOCK][EPW4][64];
    __shared__ int32_t s_pdec_all[WAVES_PER_BLOCK][64];
    __shared__ double s_cap[64], s_icap[64];
    __shared__ int4 s_rowi_all[WAVES_PER_BLOCK][64];      // observation rows to write: {home | env << 8, queue, slot, -}
    __shared__ float4 s_rowf_all[WAVES_PER_BLOCK][64];    //                            {data, workload, time left, -}

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = SC ? SC : p.S, D = DC ? DC : p.D, w = S + 4;
    const int RB = SC ? srec_bytes(SC) : p.rec_bytes;
    ESTAMP(0);
    // the capacity tables are requested first and waited for AFTER the step's own loads are in flight and the first
    // Philox round is done (one memory round trip per workgroup instead of two in a row: 1.8 k of a step's 17 k cycles)
    double cap_in = 1.0, icap_in = 1.0;
    if (threadIdx.x < 64 && (int)threadIdx.x < S) { cap_in = cc->cap[threadIdx.x]; icap_in = cc->inv_cap[threadIdx.x]; }
    const int64_t env0 = ((int64_t)blockIdx.x * WAVES_PER_BLOCK + wave) * EPW4;
    const bool wave_ok = env0 < p.n_envs;        // a wavefront past the batch shadows environment 0 up to the barrier
    ESTAMP(1);
    double (*s_load)[64] = s_load_all[wave];
    int32_t (*s_queue)[64] = s_queue_all[wave];
    float (*s_ratio)[64] = s_ratio_all[wave];
    double* s_lat = s_lat_all[wave];
    double* s_en = s_en_all[wave];

    const int g = lane >> 4, sl = lane & 15, gbase = g * GW4;
    const int64_t env = env0 + g;
    const bool valid = env < p.n_envs;
    const uint32_t eid = p.env_offset + (uint32_t)env;
    const uint32_t step_no = RESET ? 0u : (uint32_t)p.step_no;
    const int64_t renv = valid ? env : (wave_ok ? env0 : 0);     // lanes of an absent environment shadow a real one and never store
    unsigned char* rec = state + renv * (int64_t)RB;
    double* g_scal = reinterpret_cast<double*>(rec + srec_off_scal(S));
    int32_t* g_int = reinterpret_cast<int32_t*>(rec + srec_off_int(S));
    PendEntry* g_pend = reinterpret_cast<PendEntry*>(rec + srec_off_pend(S));
    constexpr int PEND_PRE = 6;      // entries 0..5 lie in the two lines that hold the scalars: requested with them, unconditionally
    const uint32_t below = (1u << sl) - 1u;

    // ---- every global load of the step is issued up front, none depends on another's result (the step is bound by
    //      memory latency at 3-5 wavefronts per SIMD): scalars, the scalar chain's row, slot sl of the pending list
    //      whether it is in use or not (the list has 64 entries, sl < 16), its per-slot action, the server state --------
    double t_now = 0.0, el_now = 0.0, acc_lat = 0.0, acc_en = 0.0;
    int k_old = 0, tasks_before = 0, n_viol_before = 0, n_done_before = 0;
    int dec = 0, tgt_in = -1;
    float res_in = 1.0f;
    double wl = 1.0, data = 1.0, ddl = 0.0;
    if constexpr (!RESET) {
        t_now = g_scal[SS_TIME]; el_now = g_scal[SS_ELAPSED]; acc_lat = g_scal[SS_SUMLAT]; acc_en = g_scal[SS_SUMEN];
        tasks_before = g_int[SI_TASKS]; n_viol_before = g_int[SI_VIOL]; n_done_before = g_int[SI_DONE];
        k_old = g_int[SI_PENDING];
        if (sl < PEND_PRE) { const PendEntry pe = g_pend[sl]; dec = pe.dec; wl = pe.wl; data = pe.data; ddl = pe.ddl; }
        if (p.by_slot) { tgt_in = placement[renv * PCAP + sl]; res_in = resource[renv * PCAP + sl]; }
    }
    const double window = chain_win[renv];
    const int n_new = chain_n[renv];
    double srv_l[EPW4];
    int srv_q[EPW4];
#pragma unroll
    for (int e = 0; e < EPW4; ++e) {
        srv_l[e] = 0.0;
        srv_q[e] = 0;
        if constexpr (!RESET) {
            if (env0 + e < p.n_envs && lane < S) {
                const unsigned char* r = state + (env0 + e) * (int64_t)RB;
                srv_l[e] = reinterpret_cast<const double*>(r)[lane];
                srv_q[e] = reinterpret_cast<const int32_t*>(r + srec_off_queue(S))[lane];
            }
        }
    }

    // ---- Philox while the loads fly: arrivals 0..7 (lanes 0-7: P_TASK_A, 8-15: P_TASK_B, exchanged with the partner
    //      lane); arrivals 8..15 take a second round if any environment of the wavefront has that many ---------------
    double a_wl = 0.0, a_data = 0.0, a_offs = 0.0;
    int a_dec = 0;
    const bool is_a = sl < 8;
    auto task_round = [&](int rd) {
        const uint32_t j = (uint32_t)(8 * rd + (sl & 7));
        const uint4 b = philox(eid, p.episode, step_no, ((is_a ? P_TASK_A : P_TASK_B) << 24) | j, p.seed_lo, p.seed_hi);
        const double mine0 = (is_a ? cc->wl_lo : cc->ddl_lo) + (is_a ? cc->wl_span : cc->ddl_span) * u53(b.x, b.y);
        const double mine1 = cc->data_lo + cc->data_span * u53(b.z, b.w);
        const int minez = (int)__umulhi(b.z, (uint32_t)D);
        const double oth0 = __shfl_xor(mine0, 8), oth1 = __shfl_xor(mine1, 8);
        const int othz = __shfl_xor(minez, 8);
        if ((sl >> 3) == rd) {       // consumer of arrival 8*rd + i is lane 8*rd + i: A-lane in round 0, B-lane in round 1
            a_wl = is_a ? mine0 : oth0;
            a_data = is_a ? mine1 : oth1;
            a_offs = is_a ? oth0 : mine0;
            a_dec = is_a ? othz : minez;
        }
    };
    ESTAMP(2);
    task_round(0);
    if (threadIdx.x < 64) { s_cap[threadIdx.x] = cap_in; s_icap[threadIdx.x] = icap_in; }
    __syncthreads();
    if (!wave_ok) return;                        // whole wavefronts leave; nothing below synchronises across wavefronts
    ESTAMP(3);

    const bool slow = valid && (k_old > GW4 || n_new > GW4);
    const bool fast = valid && !slow;
    const uint64_t slow_bits = __ballot(slow && sl == 0);
    if (__ballot(fast && n_new > 8) != 0ull) task_round(1);
#pragma unroll
    for (int e = 0; e < EPW4; ++e) { s_load[e][lane] = srv_l[e]; s_queue[e][lane] = srv_q[e]; }
    const bool have = !RESET && fast && sl < k_old;
    if constexpr (!RESET) {
        if (have && sl >= PEND_PRE) {    // more than six pending deciders (rare): their entries, now that the count is known
            const PendEntry pe = g_pend[sl];
            dec = pe.dec; wl = pe.wl; data = pe.data; ddl = pe.ddl;
        }
        if (!p.by_slot && have) { tgt_in = placement[env * (int64_t)D + dec]; res_in = resource[env * (int64_t)D + dec]; }
        if (!have) { dec = 0; tgt_in = -1; res_in = 1.0f; wl = 1.0; data = 1.0; ddl = 0.0; }
    }

    double reward = 0.0;
    int k = 0;
    ESTAMP(4);
    if constexpr (!RESET) {
        double net_bw = 0.0, net_lat = 0.0, ex_u = 2.0, ex_res = 0.0;
        int ex_tgt = 0;
        if (__ballot(have) != 0ull) {
            const uint4 b = philox(eid, p.episode, step_no, (P_NET << 24) | (uint32_t)dec, p.seed_lo, p.seed_hi);
            net_bw = cc->bw_lo + cc->bw_span * u53(b.x, b.y);
            net_lat = cc->lat_lo + cc->lat_span * u53(b.z, b.w);
            if (p.explore) {
                const uint4 e1 = philox(eid, p.episode, step_no, (P_EXPL << 24) | (uint32_t)dec, p.seed_lo, p.seed_hi);
                ex_u = u53(e1.x, e1.y);
                ex_tgt = (int)__umulhi(e1.z, (uint32_t)S);
                if (__ballot(have && !(ex_u > p.eps)) != 0ull) {
                    const uint4 e2 = philox(eid, p.episode, step_no, (P_EXPL_RES << 24) | (uint32_t)dec, p.seed_lo, p.seed_hi);
                    ex_res = 0.1 + (1.0 - 0.1) * u53(e2.x, e2.y);
                }
            }
        }
        ESTAMP(5);
        // ---- placement phase (MRRL.PY:397-438) ----------------------------------------------------------------
        int tgt = tgt_in;
        const bool acting = have && tgt >= 0 && tgt < S;
        float res = acting ? res_in : 1.0f;
        const uint32_t act_bits = (uint32_t)((__ballot(acting) >> gbase) & 0xffffull);
        const int slot = __popc(act_bits & below);
        k = __popc(act_bits);
        if (acting && p.explore && !(ex_u > p.eps)) {   // _maybe_apply_env_exploration, MRRL.PY:258-274
            tgt = ex_tgt;
            res = (float)ex_res;
        }
        const int home = dec % S;
        double t_trans = 0.0;
        if (acting && tgt != home) {         // MRRL.PY:409-414
            const double bw = net_bw * 1e6 / 8;
            t_trans = (data / bw) + net_lat;
        }
        const int tl = acting ? tgt : 0;
        const double cap_t = s_cap[tl], icap_t = s_icap[tl];
        const double alloc = cap_t * (double)res;        // MRRL.PY:418 (float64 * float32)
        const double gcyc = qdiv(alloc, 1e9, 1e-9);
        const double t_exec = wl / gcyc;                  // MRRL.PY:419
        const double energy = wl * gcyc;                  // MRRL.PY:423
        const double inc = t_exec * gcyc;                 // MRRL.PY:437
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        ESTAMP(6);
        // ordered chain over acting slots, the four environments side by side (MRRL.PY:416,437)
        int kmax = k;
        kmax = max(kmax, __shfl_xor(kmax, 16));
        kmax = max(kmax, __shfl_xor(kmax, 32));
        bool viol = false;
        double t_total = 0.0;
        // The order only matters between placements on the SAME server.  Every acting slot marks its target (the ratio
        // table is free until the arrivals phase) and reads the mark back: if nobody found another slot's number, no two
        // slots of an environment share a target and all of them take their turn at once (one LDS round trip instead of
        // one per slot: about 1.2 k of a step's 17 k cycles at 2-3 pending deciders per environment).
        int* const mark = reinterpret_cast<int*>(&s_ratio[g][0]);
        if (acting) mark[tgt] = slot;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const bool clash = acting && mark[tgt] != slot;
        const int turns = __ballot(clash) == 0ull ? (kmax > 0 ? 1 : 0) : kmax;
        for (int s = 0; s < turns; ++s) {
            if (acting && (turns < kmax || slot == s)) {
                const double l = s_load[g][tgt];
                const double t_queue = qdiv(l, cap_t, icap_t);
                t_total = t_trans + t_queue + t_exec;
                viol = (t_now + t_total) > ddl;          // MRRL.PY:431
                s_lat[gbase + slot] = t_total;
                s_en[gbase + slot] = energy;
                s_load[g][tgt] = l + inc;
                s_queue[g][tgt] += 1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        const int n_viol = __popc((uint32_t)((__ballot(viol) >> gbase) & 0xffffull));
        ESTAMP(7);

        // reward (MRRL.PY:326-338) and episode accumulators (MRRL.PY:463-466); every lane of the group computes its copy
        double sum_lat = 0.0, sum_en = 0.0;
        if (kmax < 8) {
            // fewer than eight acting slots in every environment of the wavefront: numpy's reduction is the plain left-to-right
            // sum.  The (at most seven) terms are fetched in one batch and added under a per-group mask; loops with one
            // dependent LDS read per term were 2 k of a step's 17 k cycles.
            double v[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) v[i] = s_lat[gbase + i];
            double r = -0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) { r = i < k ? r + v[i] : r; acc_lat = i < k ? acc_lat + v[i] : acc_lat; }
            sum_lat = 0.0 + r;
#pragma unroll
            for (int i = 0; i < 7; ++i) v[i] = s_en[gbase + i];
            r = -0.0;
#pragma unroll
            for (int i = 0; i < 7; ++i) { r = i < k ? r + v[i] : r; acc_en = i < k ? acc_en + v[i] : acc_en; }
            sum_en = 0.0 + r;
        } else if (k > 0) {
            sum_lat = np_sum_small(s_lat + gbase, k); sum_en = np_sum_small(s_en + gbase, k);
            for (int i = 0; i < k; ++i) { acc_lat += s_lat[gbase + i]; acc_en += s_en[gbase + i]; }
        }
        if (k > 0) {
            const double kd = (double)k, ik = cc->inv_k[k];
            const double cost = (cc->w_lat * qdiv(sum_lat, kd, ik)) + (cc->w_en * qdiv(sum_en, kd, ik)) +
                                (cc->w_viol * (double)n_viol);
            reward = -cost;
        }
        if (out.step_latency && fast) {
            double* sl_out = out.step_latency + env * PCAP;
            sl_out[sl] = sl < k ? s_lat[gbase + sl] : __builtin_nan("");
            sl_out[sl + 16] = __builtin_nan(""); sl_out[sl + 32] = __builtin_nan(""); sl_out[sl + 48] = __builtin_nan("");
        }
        if (lat_log && acting) {
            const int64_t at = (int64_t)tasks_before + slot;
            if (at < p.lat_cap) lat_log[env * p.lat_cap + at] = t_total;
        }
        tasks_before += k;
        n_viol_before += n_viol;
        n_done_before += k - n_viol;
    }

    ESTAMP(8);
    // ---- load ratios BEFORE this step's decay: server phase over the four environments -----------------------------
    {
        const double cap_me = s_cap[lane], icap_me = s_icap[lane];
#pragma unroll
        for (int e = 0; e < EPW4; ++e) s_ratio[e][lane] = lane < S ? (float)qdiv(s_load[e][lane], cap_me, icap_me) : 0.0f;
    }

    // ---- arrivals (MRRL.PY:340-383) ---------------------------------------------------------------------------
    const bool has = fast && sl < n_new;
    const double a_deadline = t_now + a_offs;              // MRRL.PY:217
    const int a_home = a_dec % S;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int q_obs = has ? s_queue[g][a_home] : 0;               // after the placements, before this step's bumps (MRRL.PY:311)
    // a later arrival of a decider that already got a task this step only bumps its home server's queue
    // (MRRL.PY:358-359); a new decider observes the bumps earlier arrivals made to ITS home server (D > S only)
    bool dup = false;
    {
        int nmax = has ? n_new : 0;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) nmax = max(nmax, __shfl_xor(nmax, m));
        for (int i = 0; i + 1 < nmax; ++i) {
            const int di = __shfl(a_dec, gbase + i), hi = __shfl(a_home, gbase + i);
            const int dupi = __shfl((int)dup, gbase + i);   // final for lane i: it depends on the iterations before i only
            if (sl > i) {
                if (a_dec == di) dup = true;
                if (dupi && hi == a_home) q_obs += 1;
            }
        }
    }
    ESTAMP(9);
    const bool is_new = has && !dup;
    const uint32_t new_bits = (uint32_t)((__ballot(is_new) >> gbase) & 0xffffull);
    const int rank = __popc(new_bits & below);
    const int n_pending = __popc(new_bits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (has && dup) atomicAdd(&s_queue[g][a_home], 1);

    // observation rows of the new pending deciders (_get_obs, MRRL.PY:307-324), compact, in arrival order
    const double a_rem = a_deadline - t_now;
    const float f_data = (float)a_data, f_wl = (float)a_wl, f_rem = (float)(a_rem > 0 ? a_rem : 0.0);
    if ((w & 3) == 0 && (reinterpret_cast<uintptr_t>(out.pend_obs) & 15) == 0) {
        // Rows of w = S + 4 floats are w / 4 aligned 16-byte pieces.  Every new decider publishes its row's parameters at
        // (rows of the preceding environments of the wavefront) + (its slot); then the wavefront writes all rows of its four
        // environments piece by piece, 64 pieces per pass (about nine rows = 153 pieces per wavefront at 64 x 256).  One row
        // at a time - five readlanes, then two passes over the 68 columns, the second with four lanes - was half of the
        // kernel's instructions.
        int4* s_rowi = s_rowi_all[wave];
        float4* s_rowf = s_rowf_all[wave];
        const int np0 = rl_i(n_pending, 0), np1 = rl_i(n_pending, 16), np2 = rl_i(n_pending, 32), np3 = rl_i(n_pending, 48);
        const int rb1 = np0, rb2 = rb1 + np1, rb3 = rb2 + np2, n_rows = rb3 + np3;
        if (is_new) {
            const int at = (g == 0 ? 0 : g == 1 ? rb1 : g == 2 ? rb2 : rb3) + rank;
            s_rowi[at] = make_int4(a_home | (g << 8), q_obs, rank, 0);
            s_rowf[at] = make_float4(f_data, f_wl, f_rem, 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int pieces = w >> 2;
        const uint32_t magic_p = (uint32_t)((0x100000000ull + (uint32_t)pieces - 1u) / (uint32_t)pieces);   // t / pieces for t < 2^16
        const int n_items = n_rows * pieces;
        for (int t = lane; t < n_items; t += 64) {
            const int rr = (int)__umulhi((uint32_t)t, magic_p), m = t - rr * pieces;
            const int4 pi = s_rowi[rr];
            const float4 pf = s_rowf[rr];
            const int h = pi.x & 255, e = pi.x >> 8;
            const float* ra = s_ratio[e];
            float v[4];
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) {
                int i = 4 * m + k2 - 5;                 // column c = 4m + k2 >= 5 shows server i, the home server left out
                i = i < 0 ? 0 : i;
                v[k2] = ra[i < h ? i : i + 1];
            }
            if (m == 0) { v[0] = ra[h]; v[1] = (float)pi.y; v[2] = pf.x; v[3] = pf.y; }
            if (m == 1) v[0] = pf.z;
            float* row = out.pend_obs + ((env0 + e) * PCAP + pi.z) * (int64_t)w;
            *reinterpret_cast<float4*>(row + 4 * m) = make_float4(v[0], v[1], v[2], v[3]);
        }
    } else {
        const uint64_t all_new = __ballot(is_new);
#pragma unroll
        for (int e = 0; e < EPW4; ++e) {
            uint32_t nb = (uint32_t)((all_new >> (16 * e)) & 0xffffull);
            int r = 0;
            while (nb != 0u) {
                const int j = 16 * e + __ffs(nb) - 1;
                nb &= nb - 1u;
                const int h = rl_i(a_home, j);
                const float c1 = (float)rl_i(q_obs, j), c2 = rl_f(f_data, j), c3 = rl_f(f_wl, j), c4 = rl_f(f_rem, j);
                float* row = out.pend_obs + ((env0 + e) * PCAP + r) * (int64_t)w;
                for (int c = lane; c < w; c += 64) {
                    float v;
                    if (c == 0) v = s_ratio[e][h];
                    else if (c == 1) v = c1;
                    else if (c == 2) v = c2;
                    else if (c == 3) v = c3;
                    else if (c == 4) v = c4;
                    else { const int i = c - 5; v = s_ratio[e][i < h ? i : i + 1]; }
                    row[c] = v;
                }
                ++r;
            }
        }
    }
    if (is_new) {                                           // MRRL.PY:361: the decider's pending task
        g_pend[rank] = PendEntry{a_wl, a_data, a_deadline, a_dec, 0};
        out.pend_decider[env * PCAP + rank] = a_dec;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    ESTAMP(10);
    // ---- decay, clock, outputs (MRRL.PY:364-375,444-459) ---------------------------------------------------------
    {
        const double cap_me = s_cap[lane];
#pragma unroll
        for (int e = 0; e < EPW4; ++e) {
            const bool fast_e = __builtin_amdgcn_readlane((int)fast, 16 * e) != 0;       // uniform
            if (!fast_e) continue;
            double l = s_load[e][lane];
            int q = s_queue[e][lane];
            if constexpr (!RESET) {
                const double win_e = rl_d(window, 16 * e);
                l = l - cap_me * win_e;
                if (!(l > 0)) l = 0.0;
                q = q - 1 > 0 ? q - 1 : 0;
            }
            if (lane < S) {
                unsigned char* r = state + (env0 + e) * (int64_t)RB;
                reinterpret_cast<double*>(r)[lane] = l;
                reinterpret_cast<int32_t*>(r + srec_off_queue(S))[lane] = q;
            }
        }
    }
    if (fast && sl == 0) {
        if constexpr (!RESET) {
            g_scal[SS_TIME] = t_now + window;
            g_scal[SS_ELAPSED] = el_now + window;
            g_scal[SS_SUMLAT] = acc_lat;
            g_scal[SS_SUMEN] = acc_en;
            g_int[SI_STEP] = (int32_t)step_no;
            g_int[SI_TASKS] = tasks_before;
            g_int[SI_VIOL] = n_viol_before;
            g_int[SI_DONE] = n_done_before;
            if (lat_log) g_int[SI_LOGGED] = (int32_t)((int64_t)tasks_before < p.lat_cap ? (int64_t)tasks_before : p.lat_cap);
            if (out.reward) out.reward[env] = reward;
            if (out.emitted) out.emitted[env] = n_pending > 0;
            if (out.reward_seen) out.reward_seen[env] = n_pending > 0 ? reward : 0.0;
            if (out.done) out.done[env] = (int32_t)step_no >= p.horizon;
        } else {
            g_scal[SS_TIME] = 0.0; g_scal[SS_ELAPSED] = 0.0; g_scal[SS_SUMLAT] = 0.0; g_scal[SS_SUMEN] = 0.0;
            g_int[SI_STEP] = 0; g_int[SI_TASKS] = 0; g_int[SI_VIOL] = 0; g_int[SI_DONE] = 0; g_int[SI_LOGGED] = 0;
        }
        g_int[SI_PENDING] = n_pending;
        out.n_pending[env] = n_pending;
    }

    ESTAMP(11);
    // ---- environments beyond the 16-slot groups: the general path, one at a time on the whole wavefront -------------
    if (slow_bits != 0ull) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint64_t sb = slow_bits;
        while (sb != 0ull) {
            const int e = (__ffsll((unsigned long long)sb) - 1) >> 4;
            sb &= sb - 1ull;
            const double win_e = rl_d(window, 16 * e);
            const int nn_e = rl_i(n_new, 16 * e);
            step_one_env<RESET, SC, DC>(p, state, cc, placement, resource, win_e, nn_e, out, lat_log, err_flag, env0 + e, lane, s_load[0],
                                s_lat, s_en, s_queue[0], s_ratio[0], s_pdec_all[wave]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// Episode metrics (MRRL.PY:470-504); p99 from the latency log when present.  One workgroup per environment.
// Episode metrics, one wavefront per environment; p99 by selection (em_env_math.h: wave_p99): logs up to 4,096 tasks in
// registers (the default scaled episode logs about 2,300-2,900: some 30 extraction rounds), larger ones streamed.
constexpr int SSEL_PER_LANE = 64;
__global__ __launch_bounds__(256) void scaled_metrics_select_kernel(const ScaledParams p, const unsigned char* __restrict__ state,
                                                                    const EnvConst* __restrict__ cc,
                                                                    const double* __restrict__ lat_log, double* __restrict__ out6)
{
    const int lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= p.n_envs) return;
    const int S = p.S;
    const unsigned char* rec = state + env * (int64_t)p.rec_bytes;
    const double* scal = reinterpret_cast<const double*>(rec + srec_off_scal(S));
    const int32_t* ri = reinterpret_cast<const int32_t*>(rec + srec_off_int(S));
    const int tasks = ri[SI_TASKS];
    const int logged = lat_log ? ri[SI_LOGGED] : 0;
    double p99 = __builtin_nan("");
    if (tasks == 0) p99 = 0.0;
    else if (lat_log && logged == tasks) p99 = wave_p99_any<SSEL_PER_LANE>(lat_log + env * p.lat_cap, logged, lane);
    if (lane == 0) {
        double* o = out6 + env * 6;
        const double cnt = (double)(tasks ? tasks : 1);
        o[0] = tasks ? scal[SS_SUMLAT] * 1000 / (double)tasks : 0.0;
        o[1] = p99;
        o[2] = tasks ? scal[SS_SUMEN] / (double)tasks : __builtin_nan("");
        o[3] = (double)ri[SI_VIOL] / cnt;
        const double el = scal[SS_ELAPSED];
        o[4] = (double)ri[SI_DONE] / (el > 1e-6 ? el : 1e-6);
        const double samples = (double)(ri[SI_STEP] + 1);
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < S; ++i) {
            const double m_i = (cc->cap[i] * el / (cc->cap[i] + 1e-9)) / samples;
            s1 += m_i;
            s2 += m_i * m_i;
        }
        o[5] = (s1 * s1) / ((double)S * s2 + 1e-9);
    }
}


}  // namespace

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct ems_handle {
    em_config cfg;
    int32_t n_deciders = 0;
    int64_t n_envs = 0;
    int device = 0;
    uint64_t seed = 0, env_offset = 0;
    uint32_t episode = 0;
    int64_t explore_counter = 0;
    int explore_off = 0;
    int64_t lat_cap = 0;
    int32_t step = 0;
    unsigned char* d_state = nullptr;
    EnvConst* d_const = nullptr;
    double* d_lat_log = nullptr;
    int32_t* d_err = nullptr;
    // scalar chain computed ahead by scaled_chain_kernel: rows [chain_first, chain_first + chain_rows) of steps
    double* d_chain_clock = nullptr;   // [n_envs] sim_time after the last computed row
    double* d_chain_win = nullptr;     // [2][CHAIN_ROWS][n_envs]: the window in use and the one computed ahead
    int32_t* d_chain_n = nullptr;      // [2][CHAIN_ROWS][n_envs]
    int32_t chain_first = 0, chain_rows = 0;
    // the NEXT window is computed on a stream of the handle's own while the current window's steps run (the chain depends on
    // the clock only): 28 us of one-lane-per-environment latency every 16 steps, 1.8 us per step, off the caller's stream
    hipStream_t chain_stream = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr};   // window in half b is complete
    hipEvent_t ev_used = nullptr;                  // the caller's stream has passed the point where half b^1 may be rewritten
    int32_t cur_half = 0, ahead_first = -1;        // half holding the current window; first step of the window computed ahead (-1: none)
    int envs_per_wave = 1;             // 4: scaled_env4_kernel, 1: scaled_env_kernel (identical results); see pick_epw
    bool was_reset = false;
};

namespace {

// Four environments per wavefront share the lane-sparse instructions.  With the rows of a wavefront written piecewise
// (scaled_env4_kernel) its one-wavefront latency is below the one-environment kernel's at every batch size (MI355X, us per
// step, four / one environment per wavefront: 12.3 / 15.0 at 4,096, 14.0 / 20.9 at 8,192, 18.0 / 26.8 at 16,384, 62.7 / 104
// at 65,536), so it is the default; ems_set_envs_per_wave(1) selects scaled_env_kernel (identical results, tested).
int pick_epw(int64_t n_envs) { (void)n_envs; return 4; }

ScaledParams make_sparams(const ems_handle* h)
{
    const em_config& c = h->cfg;
    ScaledParams p;
    std::memset(&p, 0, sizeof p);
    p.S = c.n_nodes;
    p.D = h->n_deciders;
    p.horizon = c.episode_length;
    p.rate_mode = c.arrival_rate_lo == c.arrival_rate_hi ? 3 : c.arrival_mode;
    p.explore = h->explore_off ? 0 : 1;
    p.rec_bytes = srec_bytes(p.S);
    double eps = 0.0;                                   // _get_current_epsilon, MRRL.PY:249-256
    if (!h->explore_off) {
        double frac = (double)h->explore_counter / (double)(c.decay_steps > 1 ? c.decay_steps : 1);
        if (frac > 1.0) frac = 1.0;
        eps = c.initial_epsilon - (c.initial_epsilon - c.final_epsilon) * frac;
    }
    p.eps = eps;
    p.seed_lo = (uint32_t)h->seed; p.seed_hi = (uint32_t)(h->seed >> 32);
    p.episode = h->episode;
    p.step_no = h->step + 1;
    p.env_offset = (uint32_t)h->env_offset;
    p.n_envs = h->n_envs;
    p.lat_cap = h->lat_cap;
    return p;
}

// Makes sure the scalar chain of step `step_no` (0 = reset) is in the buffers; returns its row through `row`.
int launch_chain(ems_handle* h, const ScaledParams& p, int first_step, int half, hipStream_t s)
{
    const dim3 grid((unsigned)((h->n_envs + 255) / 256)), block(256);
    const int64_t off = (int64_t)half * CHAIN_ROWS * h->n_envs;
    hipLaunchKernelGGL(scaled_chain_kernel, grid, block, 0, s, p, h->d_const, h->d_chain_clock, h->d_chain_win + off,
                       h->d_chain_n + off, first_step, (int)CHAIN_ROWS);
    return em::hip_check(hipGetLastError(), "scaled_chain_kernel launch");
}

// Row (of the two-window buffers) that holds step_no's scalar chain; computes what is missing.  Steps come in sequence
// (0 = reset, then 1, 2, ...).  The window after the current one is computed ahead on the handle's stream: it may start once
// the caller's stream has passed this point (the half it overwrites was last read by the steps of the window before the
// current one, and the clock it continues from was written by the launch above or by its own predecessor on that stream).
int ensure_chain(ems_handle* h, const ScaledParams& p, int step_no, hipStream_t s, int* row)
{
    if (step_no == 0 || step_no < h->chain_first || step_no >= h->chain_first + h->chain_rows) {
        if (step_no != 0 && step_no != h->chain_first + h->chain_rows)
            return em::fail(EM_ERR_STATE, "scalar chain out of sequence (step %d, have %d..%d)", step_no, h->chain_first,
                            h->chain_first + h->chain_rows - 1);
        if (step_no != 0 && h->ahead_first == step_no) {
            EM_HIP(hipStreamWaitEvent(s, h->ev_ready[h->cur_half ^ 1], 0));
            h->cur_half ^= 1;
        } else {
            // a reset (or a caller that skipped the look-ahead): whatever is still running ahead writes the clock and the other
            // half, so the caller's stream goes behind it before the window is computed in place
            if (h->ahead_first >= 0) EM_HIP(hipStreamWaitEvent(s, h->ev_ready[h->cur_half ^ 1], 0));
            h->cur_half = step_no == 0 ? 0 : (h->cur_half ^ 1);
            const int rc = launch_chain(h, p, step_no, h->cur_half, s);
            if (rc) return rc;
        }
        h->ahead_first = -1;
        h->chain_first = step_no;
        h->chain_rows = CHAIN_ROWS;
        const int next_first = step_no + CHAIN_ROWS;
        if (h->chain_stream && next_first <= p.horizon) {
            EM_HIP(hipEventRecord(h->ev_used, s));
            EM_HIP(hipStreamWaitEvent(h->chain_stream, h->ev_used, 0));
            const int rc = launch_chain(h, p, next_first, h->cur_half ^ 1, h->chain_stream);
            if (rc) return rc;
            EM_HIP(hipEventRecord(h->ev_ready[h->cur_half ^ 1], h->chain_stream));
            h->ahead_first = next_first;
        }
    }
    *row = h->cur_half * CHAIN_ROWS + (step_no - h->chain_first);
    return EM_OK;
}

int check_sout(const ems_step_out* out)
{
    if (!out || !out->n_pending || !out->pend_decider || !out->pend_obs)
        return em::fail(EM_ERR_INVALID, "n_pending, pend_decider and pend_obs outputs are required");
    return EM_OK;
}

}  // namespace

extern "C" {

int ems_create(const em_config* cfg, int32_t n_deciders, int64_t n_envs, int device_id, uint64_t seed, uint64_t env_offset,
               int64_t latency_log_capacity, ems_handle** out)
{
    if (!cfg || !out) return em::fail(EM_ERR_INVALID, "ems_create: null argument");
    if (cfg->n_nodes < 1 || cfg->n_nodes > EM_MAX_NODES) return em::fail(EM_ERR_INVALID, "n_nodes must be 1..%d", EM_MAX_NODES);
    if (n_deciders < 1 || n_deciders > EMS_MAX_DECIDERS) return em::fail(EM_ERR_INVALID, "n_deciders must be 1..%d", EMS_MAX_DECIDERS);
    if (n_envs < 1 || n_envs > (1ll << 31)) return em::fail(EM_ERR_INVALID, "n_envs out of range");
    if (cfg->episode_length < 1) return em::fail(EM_ERR_INVALID, "episode_length must be positive");
    if (!(cfg->min_time_step > 0) || !(cfg->max_time_step >= cfg->min_time_step))
        return em::fail(EM_ERR_INVALID, "need 0 < min_time_step <= max_time_step");
    if (latency_log_capacity < 0) return em::fail(EM_ERR_INVALID, "latency_log_capacity must be >= 0");
    for (int i = 0; i < cfg->n_nodes; ++i)
        if (!(cfg->node_cpu_capacity[i] > 0)) return em::fail(EM_ERR_INVALID, "node_cpu_capacity[%d] must be positive", i);
    EM_HIP(hipSetDevice(device_id));
    ems_handle* h = new (std::nothrow) ems_handle();
    if (!h) return em::fail(EM_ERR_INVALID, "out of host memory");
    h->cfg = *cfg;
    h->n_deciders = n_deciders;
    h->n_envs = n_envs;
    h->device = device_id;
    h->seed = seed;
    h->env_offset = env_offset;
    h->explore_off = cfg->disable_env_exploration;
    h->lat_cap = latency_log_capacity;
    h->envs_per_wave = pick_epw(n_envs);
    const size_t sbytes = (size_t)n_envs * srec_bytes(cfg->n_nodes);
    hipError_t e = hipMalloc(&h->d_state, sbytes);
    if (e == hipSuccess) e = hipMemset(h->d_state, 0, sbytes);
    if (e == hipSuccess) e = hipMalloc(&h->d_const, sizeof(EnvConst));
    if (e == hipSuccess) {
        const EnvConst k = make_const(*cfg);
        e = hipMemcpy(h->d_const, &k, sizeof k, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) e = hipMalloc(&h->d_err, sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(h->d_err, 0, sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&h->d_chain_clock, sizeof(double) * (size_t)n_envs);
    if (e == hipSuccess) e = hipMalloc(&h->d_chain_win, sizeof(double) * (size_t)n_envs * CHAIN_ROWS * 2);
    if (e == hipSuccess) e = hipMalloc(&h->d_chain_n, sizeof(int32_t) * (size_t)n_envs * CHAIN_ROWS * 2);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&h->chain_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->ev_ready[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_used, hipEventDisableTiming);
    if (e == hipSuccess && h->lat_cap > 0) {
        e = hipMalloc(&h->d_lat_log, sizeof(double) * (size_t)n_envs * (size_t)h->lat_cap);
    }
    if (e != hipSuccess) {
        ems_destroy(h);
        return em::hip_fail(e, "ems_create allocation");
    }
    *out = h;
    return EM_OK;
}

int ems_destroy(ems_handle* h)
{
    if (!h) return EM_OK;
    (void)hipSetDevice(h->device);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_const) (void)hipFree(h->d_const);
    if (h->d_lat_log) (void)hipFree(h->d_lat_log);
    if (h->d_err) (void)hipFree(h->d_err);
    if (h->chain_stream) { (void)hipStreamSynchronize(h->chain_stream); (void)hipStreamDestroy(h->chain_stream); }
    for (int i = 0; i < 2; ++i) if (h->ev_ready[i]) (void)hipEventDestroy(h->ev_ready[i]);
    if (h->ev_used) (void)hipEventDestroy(h->ev_used);
    if (h->d_chain_clock) (void)hipFree(h->d_chain_clock);
    if (h->d_chain_win) (void)hipFree(h->d_chain_win);
    if (h->d_chain_n) (void)hipFree(h->d_chain_n);
    delete h;
    return EM_OK;
}

int ems_reset(ems_handle* h, int has_seed, uint64_t seed, const ems_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "ems_reset: null handle");
    const int rc = check_sout(out);
    if (rc) return rc;
    if (has_seed) { h->seed = seed; h->episode = 0; }
    h->episode += 1;
    h->was_reset = true;
    h->step = 0;
    const ScaledParams p = make_sparams(h);
    int row = 0;
    const int crc = ensure_chain(h, p, 0, (hipStream_t)stream, &row);
    if (crc) return crc;
    const int64_t per_block = (int64_t)WAVES_PER_BLOCK * h->envs_per_wave;
    const dim3 grid((unsigned)((h->n_envs + per_block - 1) / per_block)), block(64 * WAVES_PER_BLOCK);
    if (h->envs_per_wave == 4) {
        if (p.S == 64 && p.D == 256)
            hipLaunchKernelGGL((scaled_env4_kernel<true, 64, 256>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, nullptr,
                               nullptr, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
        else
            hipLaunchKernelGGL((scaled_env4_kernel<true>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, nullptr,
                               nullptr, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
    } else {
        if (p.S == 64 && p.D == 256)
            hipLaunchKernelGGL((scaled_env_kernel<true, 64, 256>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, nullptr,
                               nullptr, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
        else
            hipLaunchKernelGGL((scaled_env_kernel<true>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, nullptr,
                               nullptr, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
    }
    return em::hip_check(hipGetLastError(), "scaled_env_kernel<reset> launch");
}

#ifdef EMS_STAMP
int ems_debug_stamps(unsigned long long* out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(ems_stamps), sizeof(unsigned long long) * 16);
}
#endif

int ems_step(ems_handle* h, const int32_t* placement, const float* resource, int actions_by_slot, const ems_step_out* out,
             void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "ems_step: null handle");
    if (!h->was_reset) return em::fail(EM_ERR_STATE, "ems_step before ems_reset");
    if (!placement || !resource) return em::fail(EM_ERR_INVALID, "ems_step: placement/resource are required");
    const int rc = check_sout(out);
    if (rc) return rc;
    ScaledParams p = make_sparams(h);
    p.by_slot = actions_by_slot ? 1 : 0;
    int row = 0;
    const int crc = ensure_chain(h, p, h->step + 1, (hipStream_t)stream, &row);
    if (crc) return crc;
    const int64_t per_block = (int64_t)WAVES_PER_BLOCK * h->envs_per_wave;
    const dim3 grid((unsigned)((h->n_envs + per_block - 1) / per_block)), block(64 * WAVES_PER_BLOCK);
    if (h->envs_per_wave == 4) {
        if (p.S == 64 && p.D == 256)
            hipLaunchKernelGGL((scaled_env4_kernel<false, 64, 256>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, placement,
                               resource, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
        else
            hipLaunchKernelGGL((scaled_env4_kernel<false>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, placement,
                               resource, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
    } else {
        if (p.S == 64 && p.D == 256)
            hipLaunchKernelGGL((scaled_env_kernel<false, 64, 256>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, placement,
                               resource, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
        else
            hipLaunchKernelGGL((scaled_env_kernel<false>), grid, block, 0, (hipStream_t)stream, p, h->d_state, h->d_const, placement,
                               resource, h->d_chain_win + (int64_t)row * h->n_envs, h->d_chain_n + (int64_t)row * h->n_envs, *out,
                               h->d_lat_log, h->d_err);
    }
    const int lrc = em::hip_check(hipGetLastError(), "scaled_env_kernel<step> launch");
    if (lrc) return lrc;
    h->step += 1;
    if (!h->explore_off) h->explore_counter += 1;   // MRRL.PY:467-468
    return EM_OK;
}

int ems_episode_metrics(ems_handle* h, double* out6, void* stream)
{
    if (!h || !out6) return em::fail(EM_ERR_INVALID, "ems_episode_metrics: null argument");
    const ScaledParams p = make_sparams(h);
    hipLaunchKernelGGL(scaled_metrics_select_kernel, dim3((unsigned)((h->n_envs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p,
                       h->d_state, h->d_const, h->d_lat_log, out6);
    return em::hip_check(hipGetLastError(), "scaled_metrics_select_kernel launch");
}

int ems_set_exploration(ems_handle* h, int disable_env_exploration, int64_t global_step_counter)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    if (global_step_counter < 0) return em::fail(EM_ERR_INVALID, "counter must be >= 0");
    h->explore_off = disable_env_exploration ? 1 : 0;
    h->explore_counter = global_step_counter;
    return EM_OK;
}

int64_t ems_get_exploration_counter(const ems_handle* h) { return h ? h->explore_counter : -1; }

int ems_set_envs_per_wave(ems_handle* h, int envs_per_wave)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    if (envs_per_wave == 0) envs_per_wave = pick_epw(h->n_envs);
    if (envs_per_wave != 1 && envs_per_wave != 4) return em::fail(EM_ERR_INVALID, "environments per wavefront must be 1 or 4");
    h->envs_per_wave = envs_per_wave;
    return EM_OK;
}
int64_t ems_record_bytes(const ems_handle* h) { return h ? srec_bytes(h->cfg.n_nodes) : -1; }

int ems_get_state(ems_handle* h, double* load, double* queue, double* sim_time, double* elapsed, int32_t* step,
                  int64_t* counts, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    const int S = h->cfg.n_nodes, rb = srec_bytes(S);
    std::vector<unsigned char> host((size_t)h->n_envs * rb);
    EM_HIP(hipStreamSynchronize((hipStream_t)stream));
    EM_HIP(hipMemcpy(host.data(), h->d_state, host.size(), hipMemcpyDeviceToHost));
    int32_t err = 0;
    EM_HIP(hipMemcpy(&err, h->d_err, sizeof err, hipMemcpyDeviceToHost));
    if (err) return em::fail(EM_ERR_INVALID, "device flagged an unrepresentable state (flags %d: 16 = more than %d pending deciders)", err, PCAP);
    for (int64_t b = 0; b < h->n_envs; ++b) {
        const unsigned char* rec = host.data() + b * rb;
        const double* ld = reinterpret_cast<const double*>(rec);
        const int32_t* q = reinterpret_cast<const int32_t*>(rec + srec_off_queue(S));
        const double* sc = reinterpret_cast<const double*>(rec + srec_off_scal(S));
        const int32_t* ri = reinterpret_cast<const int32_t*>(rec + srec_off_int(S));
        for (int i = 0; i < S; ++i) {
            if (load) load[b * S + i] = ld[i];
            if (queue) queue[b * S + i] = (double)q[i];
        }
        if (sim_time) sim_time[b] = sc[SS_TIME];
        if (elapsed) elapsed[b] = sc[SS_ELAPSED];
        if (step) step[b] = ri[SI_STEP];
        if (counts) { counts[3 * b] = ri[SI_TASKS]; counts[3 * b + 1] = ri[SI_VIOL]; counts[3 * b + 2] = ri[SI_DONE]; }
    }
    return EM_OK;
}

}  // extern "C"
