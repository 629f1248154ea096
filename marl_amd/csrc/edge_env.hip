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
#include "em_env_math.h"

#pragma clang fp contract(off)

// Optional in-kernel cycle stamps of workgroup 0 / lane 0 (tools/stamp_env.hip builds this file with
// -DEM_STAMP); compiled out of the library.
#ifdef EM_STAMP
__device__ unsigned long long em_stamps[16];
// p.pad0 = i (1..11, tools only): every worker wavefront ends at stamp i, so that a counter run attributes instructions to
// the program points up to there
int em_stamp_cut = 0;
#define STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || (threadIdx.x == blockDim.x - 64 && (i) >= 12))) em_stamps[i] = __builtin_readcyclecounter(); \
                      if ((i) >= 1 && (i) <= 11 && p.pad0 == (i)) return; } while (0)
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
// LDS bytes of one worker wavefront that holds epw environments of n nodes
__host__ __device__ constexpr int wave_bytes_of(int n, int epw)
{
    int wb = epw * ((37 * n + 56 + 15) & ~15);
    wb += (epw * n * (n + 4) * 4 + 15) & ~15;      // observation tile
    wb += 2 * epw * (n < 8 ? 8 : n) * 8;          // latencies, energies of the step
    wb += epw * 8;                                // acting-rank masks
    wb += 2 * epw * n * 4;                        // load ratios, first arrival per decider
    return (wb + 15) & ~15;
}
enum { I_STEP = 0, I_TASKS, I_VIOL, I_DONE, I_PENDING, I_LOGGED };
enum { S_TIME = 0, S_ELAPSED, S_SUMLAT, S_SUMEN };

// Per-launch scalars (kernel argument, SGPRs).
struct EnvParams {
    int32_t n, horizon, rate_mode, explore;
    int32_t rec_bytes, obs_floats, wave_bytes, pad0;
    double eps;
    int32_t step_no;
    float greedy_share;          // resource share of the greedy policy (ACT_GREEDY)
    int32_t pad2;
    // step_no: 1-based index of the step being taken (all environments of a handle move together)
    uint32_t seed_lo, seed_hi, episode, env_offset;
    int64_t n_envs, lat_cap;
};

// Extra scalars of the several-steps-per-launch form (em_step_many); the single-step kernels never read them.
struct MultiParams {
    int32_t k_steps, pad;        // consecutive steps taken by this launch
    int64_t hist_stride;         // per-step stride (in environments) of reward/emitted/done outputs: 0 = overwrite
    int64_t obs_hist_stride;     // same for obs/active/rank
    int64_t explore_counter0;    // exploration step counter at the first step of the launch (MRRL.PY:122)
};


// Arrivals 0..PAR_TASKS-1 of a step are drawn in parallel, one Philox block per lane, before the state
// record has even arrived (lanes 0..7: P_TASK_A, lanes 8..15: P_TASK_B); later arrivals (rare) are drawn
// inline.  The stream is counter addressed, so this is bit-identical to drawing them one after the other
// (oracle/edge_oracle.c does that).
enum { PAR_TASKS = 8, PRE_POIS_U = 6, EM_ENV_MAX_THREADS = 512, EM_LANES_PACKED = 1, EM_PACKED_MIN_ENVS = 32768 };
#ifndef EM_ENV_MIN_WAVES
#define EM_ENV_MIN_WAVES 6     // wavefronts per SIMD the register allocation aims for (a 96-register budget for the
                               // 16-lane kernel removes its 3 spilled registers but costs 13.2 vs 11.1 us at 16,384 environments)
#endif
#ifndef EM_ENV_MIN_WAVES_MULTI
#define EM_ENV_MIN_WAVES_MULTI 4
#endif

// bits of a wavefront ballot that belong to group g of width gw (LPE == 0: the packed layout, gw = n_nodes < 64)
template <int LPE>
__device__ inline uint64_t group_bits(uint64_t ballot, int g, int gw)
{
    if constexpr (LPE == 64) return ballot;
    else if constexpr (LPE == 0) return (ballot >> (g * gw)) & ((1ull << gw) - 1ull);
    else return (ballot >> (g * LPE)) & ((1ull << LPE) - 1ull);
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
// Workgroup layout: blockDim.x / 64 - 1 WORKER wavefronts (each owns 64/LPE environments, agents on lanes)
// plus ONE SCALAR wavefront (the last) that evaluates, one environment per lane, the per-environment
// scalar chain of _generate_tasks_and_obs: arrival rate -> time window -> Poisson arrival count
// (MRRL.PY:220-247,350-351).  That chain does not depend on the placements, so it runs beside the
// workers' placement phase and is handed over through LDS at the workgroup's single barrier; run on the
// workers it would be evaluated redundantly on all LPE lanes of every group.
// ACT selects where the step's actions come from: ACT_DICT placement/resource arrays (the dict form of
// MRRL.PY:280-281), ACT_RAW continuous vectors [n_envs][n][n+1] converted as _convert_action does
// (MRRL.PY:283-305; passed through the `resource` pointer), ACT_GREEDY the greedy baseline's policy
// (MRRL.PY:806-812): argmin of load/capacity by wavefront shuffles, fixed resource share.
enum { ACT_DICT = 0, ACT_RAW = 1, ACT_GREEDY = 2 };

// NC: the node count as a compile-time constant (0 = read it from the launch parameters): record offsets, row widths and
// lane-group arithmetic fold to immediates.  Instantiated for the default topology's n = 10.
template <int LPE, bool TAPE, bool RESET, int ACT, int NC = 0>
__global__ __launch_bounds__(EM_ENV_MAX_THREADS, EM_ENV_MIN_WAVES) void env_kernel(const EnvParams p, unsigned char* __restrict__ state,
                                                  const EnvConst* __restrict__ cc,
                                                  const int32_t* __restrict__ placement,
                                                  const float* __restrict__ resource, const em_tape tape,
                                                  const em_step_out out, double* __restrict__ lat_log,
                                                  int32_t* __restrict__ err_flag)
{
    // LPE = 16/32/64: power-of-two lane groups.  LPE = 0: PACKED groups of exactly n lanes (n = 10: six
    // environments per wavefront instead of four); lanes beyond the last whole group idle.  Identical results.
    const int n = NC ? NC : p.n, w = n + 4;
    const int RB = NC ? rec_bytes(NC) : p.rec_bytes, OF = NC ? NC * (NC + 4) : p.obs_floats;   // record bytes, observation floats
    const int GW = LPE ? LPE : n;        // lanes per environment
    const int EPW = 64 / GW;             // environments per worker wavefront
    const int WB = NC ? wave_bytes_of(NC, EPW) : p.wave_bytes;
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_work = (int)(blockDim.x >> 6) - 1;
    const bool scalar_wave = wave == n_work;
    const int64_t env_b0 = (int64_t)blockIdx.x * (n_work * EPW);   // first environment of the workgroup
    const uint32_t step_no = RESET ? 0u : (uint32_t)p.step_no;
    double* s_win = reinterpret_cast<double*>(smem + n_work * WB);   // [64] window per environment
    int32_t* s_nnew = reinterpret_cast<int32_t*>(s_win + 64);                  // [64] arrivals per environment
    // acting-agent totals of the workgroup's environments per 256-environment chunk (out.chunk_totals): two chunks at
    // most (a workgroup holds <= 64 environments), then a count of the worker wavefronts that have added theirs
    unsigned long long* s_tot = reinterpret_cast<unsigned long long*>(s_nnew + 64);   // [2]
    unsigned int* s_arrived = reinterpret_cast<unsigned int*>(s_tot + 2);             // [1] (+ pad)

    // worker coordinates (meaningless on the scalar wavefront)
    const int g_raw = lane / GW, sl = lane - g_raw * GW;
    const bool spare = g_raw >= EPW;                 // packed layout only: lanes past the last whole group
    const int g = spare ? 0 : g_raw, gbase = g * GW; // (they shadow group 0 and never pass a valid/agent guard)
    const int64_t env0 = env_b0 + (int64_t)wave * EPW;
    const int n_valid = (scalar_wave || env0 >= p.n_envs) ? 0 : (int)((p.n_envs - env0) < EPW ? (p.n_envs - env0) : EPW);
    const int64_t env = env0 + g;
    const bool valid = !spare && g < n_valid;
    const bool agent = valid && sl < n;
    const uint32_t envid = p.env_offset + (uint32_t)env;

    // LDS carve-up of a worker wavefront
    unsigned char* wbase = smem + (scalar_wave ? 0 : wave) * WB;
    unsigned char* rec = wbase + g * RB;
    double* r_load = reinterpret_cast<double*>(rec);
    double* r_wl = r_load + n;
    double* r_data = r_wl + n;
    double* r_ddl = r_data + n;
    double* r_scal = r_ddl + n;
    int32_t* r_queue = reinterpret_cast<int32_t*>(rec + rec_off_queue(n));
    int32_t* r_int = reinterpret_cast<int32_t*>(rec + rec_off_int(n));
    int8_t* r_rank = reinterpret_cast<int8_t*>(rec + rec_off_rank(n));
    const int slots = n < 8 ? 8 : n;                   // latency/energy slots per environment (>= 8: batched reads)
    int off = EPW * RB;
    float* s_obs_w = reinterpret_cast<float*>(wbase + off);
    float* s_obs = s_obs_w + g * OF;
    off += (EPW * OF * 4 + 15) & ~15;
    double* s_lat = reinterpret_cast<double*>(wbase + off) + g * slots;
    off += EPW * slots * 8;
    double* s_en = reinterpret_cast<double*>(wbase + off) + g * slots;
    off += EPW * slots * 8;
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(wbase + off) + g;   // ranks of the acting agents
    off += EPW * 8;
    float* s_ratio = reinterpret_cast<float*>(wbase + off) + g * n;
    off += EPW * n * 4;
    int32_t* s_first = reinterpret_cast<int32_t*>(wbase + off) + g * n;   // first arrival index per decider

    // values that live across the barrier
    double reward = 0.0, t_now = 0.0, cap_me = 1.0, icap_me = 1.0;
    int draws_lane = 0;
    // arrivals drawn in parallel per step: 8 with 16+ lanes per environment, n/2 in the packed layout
    const int par_philox = (LPE == 0) ? (GW / 2 < PAR_TASKS ? GW / 2 : PAR_TASKS) : PAR_TASKS;
    // pre-drawn task of lane sl < par_philox (philox mode)
    double pre_wl = 0.0, pre_data = 0.0, pre_offs = 0.0;
    int pre_dec = 0;

    if (scalar_wave) {
        // =========================== scalar wavefront: one environment per lane ===========================
        STAMP(12);
        const int64_t e = env_b0 + lane;
        const bool v = lane < n_work * EPW && e < p.n_envs;
        const uint32_t eid = p.env_offset + (uint32_t)e;
        double t0 = 0.0;
        if constexpr (!RESET) {
            if (v) t0 = *reinterpret_cast<const double*>(state + e * RB + rec_off_scal(n) + 8 * S_TIME);
        }
        double rate_u = 0.0, dt_u = 0.0;
        double pu[PRE_POIS_U] = {0.0};     // first uniforms of the Knuth loop, drawn while the clock loads
        if constexpr (!TAPE) {
            const uint4 b = philox(eid, p.episode, step_no, (P_DT << 24), p.seed_lo, p.seed_hi);
            dt_u = u53(b.x, b.y);
            rate_u = cc->rate_lo + cc->rate_span * u53(b.z, b.w);
#pragma unroll
            for (int i = 0; i < PRE_POIS_U / 2; ++i) {
                const uint4 q = philox(eid, p.episode, step_no, (P_POIS << 24) | (uint32_t)i, p.seed_lo, p.seed_hi);
                pu[2 * i] = u53(q.x, q.y);
                pu[2 * i + 1] = u53(q.z, q.w);
            }
        }
        double rate;
        if (p.rate_mode == 3) rate = cc->rate_lo;      // degenerate range (MRRL.PY:222-223)
        else if (p.rate_mode == 1) {
            const double cyc = cc->cycle, icyc = cc->inv_cycle;   // cycle already clamped to >= 1e-6 (MRRL.PY:226)
            const double phase = qdiv(exact_fmod_pos(t0, cyc, icyc), cyc, icyc);
            // tape mode never uses the rate (dt and the arrival count come pre-drawn)
            rate = cc->rate_lo + (0.5 + 0.5 * det_sin2pi(phase)) * cc->rate_span;
        } else if (p.rate_mode == 2) {
            if constexpr (TAPE) rate_u = v ? tape.rate_u[e] : 0.0;
            rate = rate_u;
        } else rate = cc->rate_mid;
        const double rate_c = rate > 1e-6 ? rate : 1e-6;

        double window;
        if constexpr (RESET) window = cc->window0;
        else {
            double raw;
            if constexpr (TAPE) raw = v ? tape.dt_raw[e] : 0.0;
            else raw = (1.0 / rate_c) * (-det_log(1.0 - dt_u));
            const double dt_min = cc->dt_min, dt_max = cc->dt_max;
            window = raw < dt_min ? dt_min : (raw > dt_max ? dt_max : raw);   // MRRL.PY:247
        }
        STAMP(13);
        int n_new = 0;
        if constexpr (TAPE) {
            n_new = v ? tape.n_arrivals[e] : 0;
            if (n_new > tape.arrival_slots || n_new < 0) { atomicOr(err_flag, 2); n_new = 0; }
        } else if (v) {
            const double limit = det_exp(-(rate_c * window));
            double prod = 1.0;
            bool going = true;
#pragma unroll
            for (int i = 0; i < PRE_POIS_U; ++i) {   // Knuth multiplication method on the pre-drawn uniforms
                prod = prod * pu[i];
                going = going && prod > limit;
                n_new += going ? 1 : 0;
            }
            if (going) {
                uint4 b = make_uint4(0, 0, 0, 0);
                for (uint32_t i = PRE_POIS_U;; ++i) {   // rare tail; ends at n_new == 4096 at the latest
                    if ((i & 1u) == 0u) b = philox(eid, p.episode, step_no, (P_POIS << 24) | (i >> 1), p.seed_lo, p.seed_hi);
                    const double u = (i & 1u) ? u53(b.z, b.w) : u53(b.x, b.y);
                    prod = prod * u;
                    if (prod > limit && n_new < 4096) ++n_new; else break;
                }
            }
        }
        s_win[lane] = window;
        s_nnew[lane] = n_new;
        if (lane < 2) s_tot[lane] = 0ull;
        if (lane == 2) *s_arrived = 0u;
        STAMP(14);
    } else {
        // =========================== worker wavefront, part 1: placements ================================
        STAMP(0);
        // global loads first: actions, capacity tables, the records
        int tgt_in = -1;
        float res_in = 1.0f;
        double ik_lane = 0.0;
        if (agent) {
            cap_me = cc->cap[sl];
            icap_me = cc->inv_cap[sl];
            ik_lane = cc->inv_k[sl + 1];
            if constexpr (!RESET && ACT == ACT_DICT) { tgt_in = placement[env * n + sl]; res_in = resource[env * n + sl]; }
            if constexpr (!RESET && ACT == ACT_RAW) {
                const float* v = resource + (env * n + sl) * (int64_t)(n + 1);
                bool all_zero = true, have_nan = false;
                float best = v[0];
                int bi = 0;
                for (int j = 0; j < n; ++j) {
                    const float x = v[j];
                    all_zero = all_zero && (fabsf(x) <= 1e-8f);              // np.allclose(logits, 0), MRRL.PY:292
                    if (!have_nan && (x != x)) { have_nan = true; bi = j; }   // np.argmax returns the first NaN
                    if (!have_nan && x > best) { best = x; bi = j; }
                }
                tgt_in = all_zero ? 0 : bi % n;                               // MRRL.PY:293-295
                double nrm = ((double)v[n] + 1.0) / 2.0;                      // MRRL.PY:297-299
                nrm = nrm < 0.0 ? 0.0 : (nrm > 1.0 ? 1.0 : nrm);
                res_in = (float)(0.1 + 0.9 * nrm);
            }
        }
        uint4* ldst = reinterpret_cast<uint4*>(wbase);
        const int nvec = n_valid * RB / 16;
        // a third vector per lane only where the wavefront's records are known to need it (it costs four registers)
        constexpr bool V2 = NC != 0 && (64 / (LPE ? LPE : (NC ? NC : 64))) * ((37 * NC + 56 + 15) & ~15) / 16 > 128;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
        if constexpr (!RESET) {
            const uint4* src = reinterpret_cast<const uint4*>(state + env0 * RB);
            if (lane < nvec) v0 = src[lane];
            if (lane + 64 < nvec) v1 = src[lane + 64];
            if constexpr (V2) { if (lane + 128 < nvec) v2 = src[lane + 128]; }   // packed, n = 10: 6 records = 162 vectors
        }

        // random numbers that depend only on (environment, step, lane): evaluated while the loads fly
        double net_bw = 0.0, net_lat = 0.0, ex_u = 2.0, ex_res = 0.0;
        int ex_tgt = 0;
        if constexpr (!TAPE) {
            if (sl < 2 * par_philox) {
                // lanes 0..par-1: P_TASK_A of arrival sl (workload, data); lanes par..2par-1: P_TASK_B of arrival sl-par
                const bool is_a = sl < par_philox;
                const uint4 b = philox(envid, p.episode, step_no,
                                       ((is_a ? P_TASK_A : P_TASK_B) << 24) | (uint32_t)(is_a ? sl : sl - par_philox),
                                       p.seed_lo, p.seed_hi);
                const double ra = (is_a ? cc->wl_lo : cc->ddl_lo) + (is_a ? cc->wl_span : cc->ddl_span) * u53(b.x, b.y);
                const double rb = cc->data_lo + cc->data_span * u53(b.z, b.w);
                const int rz = (int)__umulhi(b.z, (uint32_t)n);
                pre_wl = ra;
                pre_data = rb;
                const int partner = gbase + (is_a ? sl + par_philox : sl - par_philox);
                pre_offs = __shfl(ra, partner);           // lane j < par takes the deadline offset and the decider
                pre_dec = __shfl(rz, partner);            // of arrival j from lane j + par
            }
            if constexpr (!RESET) {
                if (agent) {
                    const uint4 b = philox(envid, p.episode, step_no, (P_NET << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                    net_bw = cc->bw_lo + cc->bw_span * u53(b.x, b.y);
                    net_lat = cc->lat_lo + cc->lat_span * u53(b.z, b.w);
                    if (p.explore) {
                        const uint4 e1 = philox(envid, p.episode, step_no, (P_EXPL << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                        ex_u = u53(e1.x, e1.y);
                        ex_tgt = (int)__umulhi(e1.z, (uint32_t)n);
                        if (!(ex_u > p.eps)) {
                            const uint4 e2 = philox(envid, p.episode, step_no, (P_EXPL_RES << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                            ex_res = 0.1 + (1.0 - 0.1) * u53(e2.x, e2.y);
                        }
                    }
                }
            }
        }

        // ---- stage in: records (zeros at reset), zeroed observation tile, per-step scratch -----------
        const int nstage = RESET ? EPW * RB / 16 : nvec;
        if (lane < nstage) ldst[lane] = v0;
        if (lane + 64 < nstage) ldst[lane + 64] = v1;
        if constexpr (V2) { if (lane + 128 < nstage) ldst[lane + 128] = v2; }
        if constexpr (RESET) {
            for (int i = lane + (V2 ? 192 : 128); i < nstage; i += 64) ldst[i] = make_uint4(0, 0, 0, 0);
        } else {
            const uint4* src = reinterpret_cast<const uint4*>(state + env0 * RB);
            for (int i = lane + (V2 ? 192 : 128); i < nstage; i += 64) ldst[i] = src[i];
        }
        {
            uint4* oz = reinterpret_cast<uint4*>(s_obs_w);
            for (int i = lane; i < ((EPW * OF * 4 + 15) >> 4); i += 64) oz[i] = make_uint4(0, 0, 0, 0);
            if (sl == 0) *s_mask = 0ull;
            if (sl < n) s_first[sl] = 0x7fffffff;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        STAMP(1);

        if constexpr (!RESET && ACT == ACT_GREEDY) {
            // target = argmin(nodes_load / cpu_caps), first minimum (np.argmin), the same for every acting agent
            double ratio = agent ? qdiv(r_load[sl], cap_me, icap_me) : __builtin_inf();
            int idx = sl;
            if constexpr (LPE == 0) {
                const double mine = ratio;
                for (int i = 0; i < GW; ++i) {       // packed groups are not a power of two: linear scan by shuffles
                    const double o = __shfl(mine, gbase + i);
                    if (o < ratio || (o == ratio && i < idx)) { ratio = o; idx = i; }
                }
            } else {
#pragma unroll
                for (int m = LPE / 2; m > 0; m >>= 1) {
                    const double o = __shfl_xor(ratio, m);
                    const int oi = __shfl_xor(idx, m);
                    if (o < ratio || (o == ratio && oi < idx)) { ratio = o; idx = oi; }
                }
            }
            tgt_in = agent ? idx : -1;
            res_in = p.greedy_share;
        }
        if constexpr (!RESET) {
            t_now = r_scal[S_TIME];
            // ---- placement phase: per-agent quantities in parallel on the lanes (MRRL.PY:397-423) ----
            const int rank = agent ? (int)r_rank[sl] : -1;
            int tgt = tgt_in;
            const bool acting = rank >= 0 && tgt >= 0 && tgt < n;
            float res = acting ? res_in : 1.0f;
            const double wl = acting ? r_wl[sl] : 1.0;
            const double data = acting ? r_data[sl] : 1.0;
            const double ddl = acting ? r_ddl[sl] : 0.0;
            if (acting) atomicOr(s_mask, 1ull << rank);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long rmask = *s_mask;
            // acting slot = number of acting agents that arrived earlier; k = acting agents of the environment
            const int slot = acting ? __popcll(rmask & ((1ull << rank) - 1ull)) : -1;
            const int k = valid ? __popcll(rmask) : 0;

            if (p.explore && acting) {   // _maybe_apply_env_exploration, MRRL.PY:258-274
                if constexpr (TAPE) ex_u = tape.explore_u[env * n + slot];
                draws_lane += 1;
                if (!(ex_u > p.eps)) {
                    if constexpr (TAPE) {
                        ex_tgt = tape.explore_tgt[env * n + slot];
                        ex_res = tape.explore_res[env * n + slot];
                        if (ex_tgt < 0 || ex_tgt >= n) { atomicOr(err_flag, 1); ex_tgt = 0; }
                    }
                    tgt = ex_tgt;
                    res = (float)ex_res;
                    draws_lane += 2;
                }
            }

            double t_trans = 0.0;
            if (acting && tgt != sl) {   // MRRL.PY:409-414
                if constexpr (TAPE) { net_bw = tape.bw_u[env * n + slot]; net_lat = tape.net_lat[env * n + slot]; }
                draws_lane += 2;
                const double bw = net_bw * 1e6 / 8;
                t_trans = (data / bw) + net_lat;
            }
            // capacity of the target server: held by lane `tgt` of this group
            const int tl = gbase + (acting ? tgt : sl);
            const double cap_t = __shfl(cap_me, tl), icap_t = __shfl(icap_me, tl);
            const double alloc = cap_t * (double)res;        // MRRL.PY:418 (float64 * float32)
            const double gcyc = qdiv(alloc, 1e9, 1e-9);
            const double t_exec = wl / gcyc;                  // MRRL.PY:419
            const double energy = wl * gcyc;                  // MRRL.PY:423
            const double inc = t_exec * gcyc;                 // MRRL.PY:437

            STAMP(2);
            // ---- ordered chain over acting slots: queueing delay sees earlier placements (MRRL.PY:416,437)
            bool viol = false;
            for (int s = 0; __ballot(acting && slot >= s) != 0ull; ++s) {
                if (acting && slot == s) {
                    const double l = r_load[tgt];
                    const double t_queue = qdiv(l, cap_t, icap_t);
                    const double t_total = t_trans + t_queue + t_exec;
                    viol = (t_now + t_total) > ddl;          // MRRL.PY:431
                    s_lat[s] = t_total;
                    s_en[s] = energy;
                    r_load[tgt] = l + inc;
                    r_queue[tgt] += 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const int n_viol = __popcll(group_bits<LPE>(__ballot(viol), g, GW));

            STAMP(3);
            // ---- reward (MRRL.PY:326-338) and episode accumulators (MRRL.PY:463-466) ----------------
            const int tasks_before = r_int[I_TASKS];
            double sum_lat, sum_en, acc_lat = r_scal[S_SUMLAT], acc_en = r_scal[S_SUMEN];
            if (__ballot(k > 4) == 0ull) {
                // common case: one batch of LDS reads, sums in numpy's order for fewer than 8 addends
                const double a0 = s_lat[0], a1 = s_lat[1], a2 = s_lat[2], a3 = s_lat[3];
                const double e0 = s_en[0], e1 = s_en[1], e2 = s_en[2], e3 = s_en[3];
                double rl = -0.0, re = -0.0;
                if (k > 0) { rl += a0; re += e0; acc_lat += a0; acc_en += e0; }
                if (k > 1) { rl += a1; re += e1; acc_lat += a1; acc_en += e1; }
                if (k > 2) { rl += a2; re += e2; acc_lat += a2; acc_en += e2; }
                if (k > 3) { rl += a3; re += e3; acc_lat += a3; acc_en += e3; }
                sum_lat = 0.0 + rl;
                sum_en = 0.0 + re;
            } else {
                sum_lat = np_sum_small(s_lat, k);
                sum_en = np_sum_small(s_en, k);
                for (int i = 0; i < k; ++i) { acc_lat += s_lat[i]; acc_en += s_en[i]; }
            }
            if (k > 0) {
                const double kd = (double)k, ik = __shfl(ik_lane, gbase + k - 1);
                const double cost = (cc->w_lat * qdiv(sum_lat, kd, ik)) + (cc->w_en * qdiv(sum_en, kd, ik)) +
                                    (cc->w_viol * (double)n_viol);
                reward = -cost;
            }
            if (agent) {
                if (out.step_latency) out.step_latency[env * n + sl] = sl < k ? s_lat[sl] : __builtin_nan("");
                if (lat_log && sl < k) {
                    const int64_t at = (int64_t)tasks_before + sl;
                    if (at < p.lat_cap) lat_log[env * p.lat_cap + at] = s_lat[sl];
                }
            }
            if (valid && sl == 0) {
                r_scal[S_SUMLAT] = acc_lat;
                r_scal[S_SUMEN] = acc_en;
                r_int[I_TASKS] = tasks_before + k;
                r_int[I_VIOL] += n_viol;
                r_int[I_DONE] += k - n_viol;
                if (lat_log) {
                    const int64_t end = (int64_t)tasks_before + k;
                    r_int[I_LOGGED] = (int32_t)(end < p.lat_cap ? end : p.lat_cap);
                }
            }
        }
        STAMP(4);
    }

    __syncthreads();   // the scalar wavefront's windows and arrival counts are in LDS
    if (scalar_wave) return;

    // =============================== worker wavefront, part 2: arrivals (MRRL.PY:340-383) ===============
    STAMP(5);
    const double window = s_win[wave * EPW + g];
    const int n_new = valid ? s_nnew[wave * EPW + g] : 0;
    float ratio_me = 0.0f;
    if (agent) {
        r_rank[sl] = -1;
        ratio_me = (float)qdiv(r_load[sl], cap_me, icap_me);   // load/cap BEFORE this step's decay
        s_ratio[sl] = ratio_me;
    }
    // arrivals 0..PAR-1 in parallel, arrival j on lane j of the group
    const int PAR = TAPE ? (GW < 16 ? GW : 16) : par_philox;
    const bool has = valid && sl < PAR && sl < n_new;
    double a_wl = pre_wl, a_data = pre_data, a_offs = pre_offs;
    int a_dec = pre_dec;
    if constexpr (TAPE) {
        if (has) {
            const int64_t o = env * tape.arrival_slots + sl;
            a_wl = tape.task_workload[o]; a_data = tape.task_data[o];
            a_offs = tape.task_deadline_offset[o]; a_dec = tape.task_decider[o];
            if (a_dec < 0 || a_dec >= n) { atomicOr(err_flag, 4); a_dec = 0; }
        }
    }
    const double a_deadline = t_now + a_offs;              // MRRL.PY:217
    if (has) atomicMin(&s_first[a_dec], sl);               // first arrival of each decider (MRRL.PY:358-362)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool is_new = has && s_first[a_dec] == sl;
    const uint32_t new_bits = (uint32_t)group_bits<LPE>(__ballot(is_new), g, GW);
    const int first_me = agent ? s_first[sl] : 0x7fffffff;
    const int q_me = agent ? r_queue[sl] : 0;              // queue length BEFORE this call's bumps (MRRL.PY:311)
    const bool pend_me = first_me != 0x7fffffff;
    if (is_new) {                                          // MRRL.PY:361 and the task columns of _get_obs
        r_rank[a_dec] = (int8_t)__popc(new_bits & ((1u << sl) - 1u));
        r_wl[a_dec] = a_wl; r_data[a_dec] = a_data; r_ddl[a_dec] = a_deadline;
        const double rem = a_deadline - t_now;
        float* row = s_obs + a_dec * w;
        row[2] = (float)a_data;
        row[3] = (float)a_wl;
        row[4] = (float)(rem > 0 ? rem : 0.0);
    }
    if (pend_me) { s_obs[sl * w] = ratio_me; s_obs[sl * w + 1] = (float)q_me; }
    // the other servers' load ratios: every lane writes its own ratio into the rows of the pending agents
    {
        uint64_t pm = group_bits<LPE>(__ballot(pend_me), g, GW);
        while (__ballot(pm != 0ull) != 0ull) {
            if (pm != 0ull) {
                const int r = __ffsll((unsigned long long)pm) - 1;
                pm &= pm - 1ull;
                if (agent && sl != r) s_obs[r * w + (sl < r ? sl + 5 : sl + 4)] = ratio_me;
            }
        }
    }
    if (has && !is_new) atomicAdd(&r_queue[a_dec], 1);    // MRRL.PY:359
    int n_pending = __popc(new_bits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    STAMP(6);
    if (__ballot(n_new > PAR) != 0ull) {
        // rare tail: arrivals beyond the parallel batch, one at a time in arrival order
        for (int j = PAR;; ++j) {
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
                } else {
                    const uint4 a = philox(envid, p.episode, step_no, (P_TASK_A << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                    const uint4 b = philox(envid, p.episode, step_no, (P_TASK_B << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                    wl = cc->wl_lo + cc->wl_span * u53(a.x, a.y);
                    data = cc->data_lo + cc->data_span * u53(a.z, a.w);
                    offs = cc->ddl_lo + cc->ddl_span * u53(b.x, b.y);
                    dec = (int)__umulhi(b.z, (uint32_t)n);
                }
                const double deadline = t_now + offs;          // MRRL.PY:217
                const bool fresh = r_rank[dec] < 0;            // broadcast read, before any lane updates
                const float q_now = (float)r_queue[dec];
                if (sl == 0) {
                    if (fresh) {                                // MRRL.PY:361
                        r_rank[dec] = (int8_t)n_pending;
                        r_wl[dec] = wl; r_data[dec] = data; r_ddl[dec] = deadline;
                    } else r_queue[dec] += 1;                  // MRRL.PY:359
                }
                if (fresh) {                                    // _get_obs, MRRL.PY:307-324
                    const double rem = deadline - t_now;
                    for (int c = sl; c < w; c += GW) {
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
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
    const int rank_out = agent ? (int)r_rank[sl] : -1;
    const uint64_t act_bits = group_bits<LPE>(__ballot(rank_out >= 0), g, GW);
    if (out.chunk_totals && valid && sl == 0 && n_pending > 0) {
        // {acting agents | environments with at least one << 32} of the next step, per chunk of 256 environments
        atomicAdd(&s_tot[(env >> 8) != (env_b0 >> 8) ? 1 : 0], (unsigned long long)n_pending | (1ull << 32));
    }
    if (out.n_draws) {
        int d = draws_lane;
        if constexpr (LPE == 0) {
            const int mine = d;
            d = 0;
            for (int i = 0; i < GW; ++i) d += __shfl(mine, gbase + i);
        } else {
            for (int m = LPE / 2; m > 0; m >>= 1) d += __shfl_xor(d, m, LPE);
        }
        // once per environment: rate (random mode), window (step only), arrival count, 4 per arrival
        const int draws_env = (p.rate_mode == 2 ? 1 : 0) + (RESET ? 0 : 1) + 1 + 4 * n_new;
        if (valid && sl == 0) out.n_draws[env] = d + draws_env;
    }
    if (valid && sl == 0) {
        if constexpr (!RESET) {
            r_scal[S_TIME] = t_now + window;               // delta_t > 0 always: dt_min > 0
            r_scal[S_ELAPSED] = r_scal[S_ELAPSED] + window;
            r_int[I_STEP] = (int32_t)step_no;
            if (out.reward) out.reward[env] = reward;
            if (out.emitted) out.emitted[env] = n_pending > 0;
            if (out.reward_seen) out.reward_seen[env] = n_pending > 0 ? reward : 0.0;
            if (out.done) out.done[env] = (int32_t)step_no >= p.horizon;
        }
        r_int[I_PENDING] = n_pending;
        out.active[env] = act_bits;
    }
    if (agent && out.rank) out.rank[env * n + sl] = (int8_t)rank_out;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    STAMP(10);
    // ---- stage out: records and observation tile, 16-byte lane-linear stores -------------------------
    {
        const uint4* src = reinterpret_cast<const uint4*>(wbase);
        uint4* dst = reinterpret_cast<uint4*>(state + env0 * RB);
        for (int i = lane; i < n_valid * RB / 16; i += 64) dst[i] = src[i];
        const int64_t fbase = env0 * OF;
        const int nfl = n_valid * OF;
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
    if (out.chunk_totals) {
        // the worker wavefront that arrives last adds the workgroup's sums to the global totals: two atomics per workgroup
        // (the LDS adds above were ordered before the fence/wave barrier that precedes the stage-out)
        unsigned int before = 0u;
        if (lane == 0) before = atomicAdd(s_arrived, 1u);
        before = __builtin_amdgcn_readfirstlane(before);
        if (before == (unsigned int)(n_work - 1) && lane < 2) {
            const unsigned long long v = s_tot[lane];
            if (v) atomicAdd(reinterpret_cast<unsigned long long*>(out.chunk_totals) + (env_b0 >> 8) + lane, v);
        }
    }
    STAMP(11);
}

// The several-steps-per-launch form of env_kernel (em_step_many): the same step, inside a loop over the steps of the
// launch.  The records stay in LDS between the steps and the scalar wavefront, which owns the environments' clocks,
// runs one step ahead of the workers through a two-deep buffer.  It is a separate kernel on purpose: folding the
// loop into env_kernel cost the single-step launch 15-20 % (longer live ranges, spills), and this one gets the larger
// register budget its loop-carried values need.  TAPE/RESET/MULTI are kept as constants so the two bodies stay
// comparable line by line; any change to the step arithmetic must be made in both.
template <int LPE, int ACT, bool TAPE = false, bool RESET = false, bool MULTI = true>
__global__ __launch_bounds__(EM_ENV_MAX_THREADS, EM_ENV_MIN_WAVES_MULTI) void env_multi_kernel(const EnvParams p, const MultiParams mp, unsigned char* __restrict__ state,
                                                  const EnvConst* __restrict__ cc,
                                                  const int32_t* __restrict__ placement,
                                                  const float* __restrict__ resource, const em_tape tape,
                                                  const em_step_out out, double* __restrict__ lat_log,
                                                  int32_t* __restrict__ err_flag)
{
    // LPE = 16/32/64: power-of-two lane groups.  LPE = 0: PACKED groups of exactly n lanes (n = 10: six
    // environments per wavefront instead of four); lanes beyond the last whole group idle.  Identical results.
    const int GW = LPE ? LPE : p.n;      // lanes per environment
    const int EPW = 64 / GW;             // environments per worker wavefront
    extern __shared__ __align__(16) unsigned char smem[];

    const int n = p.n, w = n + 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n_work = (int)(blockDim.x >> 6) - 1;
    const bool scalar_wave = wave == n_work;
    const int64_t env_b0 = (int64_t)blockIdx.x * (n_work * EPW);   // first environment of the workgroup
    const uint32_t step_first = RESET ? 0u : (uint32_t)p.step_no;
    const int K = MULTI ? mp.k_steps : 1;           // consecutive steps of this launch; the records stay in LDS
    const int64_t hist_stride = MULTI ? mp.hist_stride : 0, obs_hist_stride = MULTI ? mp.obs_hist_stride : 0;
    double* s_win = reinterpret_cast<double*>(smem + n_work * p.wave_bytes);   // [2][64] window per environment
    int32_t* s_nnew = reinterpret_cast<int32_t*>(s_win + 128);                 // [2][64] arrivals per environment

    // worker coordinates (meaningless on the scalar wavefront)
    const int g_raw = lane / GW, sl = lane - g_raw * GW;
    const bool spare = g_raw >= EPW;                 // packed layout only: lanes past the last whole group
    const int g = spare ? 0 : g_raw, gbase = g * GW; // (they shadow group 0 and never pass a valid/agent guard)
    const int64_t env0 = env_b0 + (int64_t)wave * EPW;
    const int n_valid = (scalar_wave || env0 >= p.n_envs) ? 0 : (int)((p.n_envs - env0) < EPW ? (p.n_envs - env0) : EPW);
    const int64_t env = env0 + g;
    const bool valid = !spare && g < n_valid;
    const bool agent = valid && sl < n;
    const uint32_t envid = p.env_offset + (uint32_t)env;

    // LDS carve-up of a worker wavefront
    unsigned char* wbase = smem + (scalar_wave ? 0 : wave) * p.wave_bytes;
    unsigned char* rec = wbase + g * p.rec_bytes;
    double* r_load = reinterpret_cast<double*>(rec);
    double* r_wl = r_load + n;
    double* r_data = r_wl + n;
    double* r_ddl = r_data + n;
    double* r_scal = r_ddl + n;
    int32_t* r_queue = reinterpret_cast<int32_t*>(rec + rec_off_queue(n));
    int32_t* r_int = reinterpret_cast<int32_t*>(rec + rec_off_int(n));
    int8_t* r_rank = reinterpret_cast<int8_t*>(rec + rec_off_rank(n));
    const int slots = n < 8 ? 8 : n;                   // latency/energy slots per environment (>= 8: batched reads)
    int off = EPW * p.rec_bytes;
    float* s_obs_w = reinterpret_cast<float*>(wbase + off);
    float* s_obs = s_obs_w + g * p.obs_floats;
    off += (EPW * p.obs_floats * 4 + 15) & ~15;
    double* s_lat = reinterpret_cast<double*>(wbase + off) + g * slots;
    off += EPW * slots * 8;
    double* s_en = reinterpret_cast<double*>(wbase + off) + g * slots;
    off += EPW * slots * 8;
    unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(wbase + off) + g;   // ranks of the acting agents
    off += EPW * 8;
    float* s_ratio = reinterpret_cast<float*>(wbase + off) + g * n;
    off += EPW * n * 4;
    int32_t* s_first = reinterpret_cast<int32_t*>(wbase + off) + g * n;   // first arrival index per decider

    // =========================== scalar wavefront: one environment per lane ===========================
    // It owns each environment's clock for the launch (t(k+1) = t(k) + window(k), the very sum the workers store),
    // so with several steps per launch it runs one step ahead of the workers; results go to a two-deep LDS buffer.
    double sc_t = 0.0;
    auto scalar_step = [&](uint32_t step_no, int buf) {
        const int64_t e = env_b0 + lane;
        const bool v = lane < n_work * EPW && e < p.n_envs;
        const uint32_t eid = p.env_offset + (uint32_t)e;
        const double t0 = sc_t;
        double rate_u = 0.0, dt_u = 0.0;
        double pu[PRE_POIS_U] = {0.0};     // first uniforms of the Knuth loop, drawn while the clock loads
        if constexpr (!TAPE) {
            const uint4 b = philox(eid, p.episode, step_no, (P_DT << 24), p.seed_lo, p.seed_hi);
            dt_u = u53(b.x, b.y);
            rate_u = cc->rate_lo + cc->rate_span * u53(b.z, b.w);
#pragma unroll
            for (int i = 0; i < PRE_POIS_U / 2; ++i) {
                const uint4 q = philox(eid, p.episode, step_no, (P_POIS << 24) | (uint32_t)i, p.seed_lo, p.seed_hi);
                pu[2 * i] = u53(q.x, q.y);
                pu[2 * i + 1] = u53(q.z, q.w);
            }
        }
        double rate;
        if (p.rate_mode == 3) rate = cc->rate_lo;      // degenerate range (MRRL.PY:222-223)
        else if (p.rate_mode == 1) {
            const double cyc = cc->cycle, icyc = cc->inv_cycle;   // cycle already clamped to >= 1e-6 (MRRL.PY:226)
            const double phase = qdiv(exact_fmod_pos(t0, cyc, icyc), cyc, icyc);
            // tape mode never uses the rate (dt and the arrival count come pre-drawn)
            rate = cc->rate_lo + (0.5 + 0.5 * det_sin2pi(phase)) * cc->rate_span;
        } else if (p.rate_mode == 2) {
            if constexpr (TAPE) rate_u = v ? tape.rate_u[e] : 0.0;
            rate = rate_u;
        } else rate = cc->rate_mid;
        const double rate_c = rate > 1e-6 ? rate : 1e-6;

        double window;
        if constexpr (RESET) window = cc->window0;
        else {
            double raw;
            if constexpr (TAPE) raw = v ? tape.dt_raw[e] : 0.0;
            else raw = (1.0 / rate_c) * (-det_log(1.0 - dt_u));
            const double dt_min = cc->dt_min, dt_max = cc->dt_max;
            window = raw < dt_min ? dt_min : (raw > dt_max ? dt_max : raw);   // MRRL.PY:247
        }
        STAMP(13);
        int n_new = 0;
        if constexpr (TAPE) {
            n_new = v ? tape.n_arrivals[e] : 0;
            if (n_new > tape.arrival_slots || n_new < 0) { atomicOr(err_flag, 2); n_new = 0; }
        } else if (v) {
            const double limit = det_exp(-(rate_c * window));
            double prod = 1.0;
            bool going = true;
#pragma unroll
            for (int i = 0; i < PRE_POIS_U; ++i) {   // Knuth multiplication method on the pre-drawn uniforms
                prod = prod * pu[i];
                going = going && prod > limit;
                n_new += going ? 1 : 0;
            }
            if (going) {
                uint4 b = make_uint4(0, 0, 0, 0);
                for (uint32_t i = PRE_POIS_U;; ++i) {   // rare tail; ends at n_new == 4096 at the latest
                    if ((i & 1u) == 0u) b = philox(eid, p.episode, step_no, (P_POIS << 24) | (i >> 1), p.seed_lo, p.seed_hi);
                    const double u = (i & 1u) ? u53(b.z, b.w) : u53(b.x, b.y);
                    prod = prod * u;
                    if (prod > limit && n_new < 4096) ++n_new; else break;
                }
            }
        }
        s_win[buf * 64 + lane] = window;
        s_nnew[buf * 64 + lane] = n_new;
        sc_t = t0 + window;          // the clock of the next step (workers add the same window, MRRL.PY:444-447)
        STAMP(14);
    };

    // worker values that persist over the launch
    double cap_me = 1.0, icap_me = 1.0, ik_lane = 0.0;
    // arrivals drawn in parallel per step: 8 with 16+ lanes per environment, n/2 in the packed layout
    const int par_philox = (LPE == 0) ? (GW / 2 < PAR_TASKS ? GW / 2 : PAR_TASKS) : PAR_TASKS;

    for (int ks = 0; ks < K; ++ks) {
    const uint32_t step_no = step_first + (uint32_t)ks;
    // values that live across the barrier
    double reward = 0.0, t_now = 0.0;
    int draws_lane = 0;
    // pre-drawn task of lane sl < par_philox (philox mode)
    double pre_wl = 0.0, pre_data = 0.0, pre_offs = 0.0;
    int pre_dec = 0;
    // epsilon of this step, _get_current_epsilon (MRRL.PY:249-256)
    double eps_k = p.eps;            // single step: evaluated on the host (same float64 formula)
    if constexpr (MULTI) {
        if (p.explore) {
            double frac = (double)(mp.explore_counter0 + ks) / cc->eps_span;
            if (frac > 1.0) frac = 1.0;
            eps_k = cc->eps_init - (cc->eps_init - cc->eps_final) * frac;
        }
    }

    if (scalar_wave) {
        if (ks == 0) {
            STAMP(12);
            if constexpr (!RESET) {
                const int64_t e = env_b0 + lane;
                if (lane < n_work * EPW && e < p.n_envs)
                    sc_t = *reinterpret_cast<const double*>(state + e * p.rec_bytes + rec_off_scal(n) + 8 * S_TIME);
            }
            scalar_step(step_first, 0);
        }
    } else {
        // =========================== worker wavefront, part 1: placements ================================
        STAMP(0);
        // global loads first: actions, capacity tables, the records
        int tgt_in = -1;
        float res_in = 1.0f;
        const int64_t aenv = MULTI ? env + (int64_t)ks * p.n_envs : env;   // row of this step in [k_steps][n_envs] action arrays
        if (agent) {
            if (ks == 0) {
                cap_me = cc->cap[sl];
                icap_me = cc->inv_cap[sl];
                ik_lane = cc->inv_k[sl + 1];
            }
            if constexpr (!RESET && ACT == ACT_DICT) { tgt_in = placement[aenv * n + sl]; res_in = resource[aenv * n + sl]; }
            if constexpr (!RESET && ACT == ACT_RAW) {
                const float* v = resource + (aenv * n + sl) * (int64_t)(n + 1);
                bool all_zero = true, have_nan = false;
                float best = v[0];
                int bi = 0;
                for (int j = 0; j < n; ++j) {
                    const float x = v[j];
                    all_zero = all_zero && (fabsf(x) <= 1e-8f);              // np.allclose(logits, 0), MRRL.PY:292
                    if (!have_nan && (x != x)) { have_nan = true; bi = j; }   // np.argmax returns the first NaN
                    if (!have_nan && x > best) { best = x; bi = j; }
                }
                tgt_in = all_zero ? 0 : bi % n;                               // MRRL.PY:293-295
                double nrm = ((double)v[n] + 1.0) / 2.0;                      // MRRL.PY:297-299
                nrm = nrm < 0.0 ? 0.0 : (nrm > 1.0 ? 1.0 : nrm);
                res_in = (float)(0.1 + 0.9 * nrm);
            }
        }
        uint4* ldst = reinterpret_cast<uint4*>(wbase);
        const int nvec = n_valid * p.rec_bytes / 16;
        uint4 v0 = make_uint4(0, 0, 0, 0), v1 = make_uint4(0, 0, 0, 0);
        if constexpr (!RESET) {
            if (ks == 0) {       // later steps of the launch find the records in LDS
                const uint4* src = reinterpret_cast<const uint4*>(state + env0 * p.rec_bytes);
                if (lane < nvec) v0 = src[lane];
                if (lane + 64 < nvec) v1 = src[lane + 64];
            }
        }

        // random numbers that depend only on (environment, step, lane): evaluated while the loads fly
        double net_bw = 0.0, net_lat = 0.0, ex_u = 2.0, ex_res = 0.0;
        int ex_tgt = 0;
        if constexpr (!TAPE) {
            if (sl < 2 * par_philox) {
                // lanes 0..par-1: P_TASK_A of arrival sl (workload, data); lanes par..2par-1: P_TASK_B of arrival sl-par
                const bool is_a = sl < par_philox;
                const uint4 b = philox(envid, p.episode, step_no,
                                       ((is_a ? P_TASK_A : P_TASK_B) << 24) | (uint32_t)(is_a ? sl : sl - par_philox),
                                       p.seed_lo, p.seed_hi);
                const double ra = (is_a ? cc->wl_lo : cc->ddl_lo) + (is_a ? cc->wl_span : cc->ddl_span) * u53(b.x, b.y);
                const double rb = cc->data_lo + cc->data_span * u53(b.z, b.w);
                const int rz = (int)__umulhi(b.z, (uint32_t)n);
                pre_wl = ra;
                pre_data = rb;
                const int partner = gbase + (is_a ? sl + par_philox : sl - par_philox);
                pre_offs = __shfl(ra, partner);           // lane j < par takes the deadline offset and the decider
                pre_dec = __shfl(rz, partner);            // of arrival j from lane j + par
            }
            if constexpr (!RESET) {
                if (agent) {
                    const uint4 b = philox(envid, p.episode, step_no, (P_NET << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                    net_bw = cc->bw_lo + cc->bw_span * u53(b.x, b.y);
                    net_lat = cc->lat_lo + cc->lat_span * u53(b.z, b.w);
                    if (p.explore) {
                        const uint4 e1 = philox(envid, p.episode, step_no, (P_EXPL << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                        ex_u = u53(e1.x, e1.y);
                        ex_tgt = (int)__umulhi(e1.z, (uint32_t)n);
                        if (!(ex_u > eps_k)) {
                            const uint4 e2 = philox(envid, p.episode, step_no, (P_EXPL_RES << 24) | (uint32_t)sl, p.seed_lo, p.seed_hi);
                            ex_res = 0.1 + (1.0 - 0.1) * u53(e2.x, e2.y);
                        }
                    }
                }
            }
        }

        // ---- stage in: records (zeros at reset), zeroed observation tile, per-step scratch -----------
        if (ks == 0) {
            const int nstage = RESET ? EPW * p.rec_bytes / 16 : nvec;
            if (lane < nstage) ldst[lane] = v0;
            if (lane + 64 < nstage) ldst[lane + 64] = v1;
            if constexpr (RESET) {
                for (int i = lane + 128; i < nstage; i += 64) ldst[i] = make_uint4(0, 0, 0, 0);
            } else {
                const uint4* src = reinterpret_cast<const uint4*>(state + env0 * p.rec_bytes);
                for (int i = lane + 128; i < nstage; i += 64) ldst[i] = src[i];
            }
        }
        {
            uint4* oz = reinterpret_cast<uint4*>(s_obs_w);
            for (int i = lane; i < ((EPW * p.obs_floats * 4 + 15) >> 4); i += 64) oz[i] = make_uint4(0, 0, 0, 0);
            if (sl == 0) *s_mask = 0ull;
            if (sl < n) s_first[sl] = 0x7fffffff;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        STAMP(1);

        if constexpr (!RESET && ACT == ACT_GREEDY) {
            // target = argmin(nodes_load / cpu_caps), first minimum (np.argmin), the same for every acting agent
            double ratio = agent ? qdiv(r_load[sl], cap_me, icap_me) : __builtin_inf();
            int idx = sl;
            if constexpr (LPE == 0) {
                const double mine = ratio;
                for (int i = 0; i < GW; ++i) {       // packed groups are not a power of two: linear scan by shuffles
                    const double o = __shfl(mine, gbase + i);
                    if (o < ratio || (o == ratio && i < idx)) { ratio = o; idx = i; }
                }
            } else {
#pragma unroll
                for (int m = LPE / 2; m > 0; m >>= 1) {
                    const double o = __shfl_xor(ratio, m);
                    const int oi = __shfl_xor(idx, m);
                    if (o < ratio || (o == ratio && oi < idx)) { ratio = o; idx = oi; }
                }
            }
            tgt_in = agent ? idx : -1;
            res_in = p.greedy_share;
        }
        if constexpr (!RESET) {
            t_now = r_scal[S_TIME];
            // ---- placement phase: per-agent quantities in parallel on the lanes (MRRL.PY:397-423) ----
            const int rank = agent ? (int)r_rank[sl] : -1;
            int tgt = tgt_in;
            const bool acting = rank >= 0 && tgt >= 0 && tgt < n;
            float res = acting ? res_in : 1.0f;
            const double wl = acting ? r_wl[sl] : 1.0;
            const double data = acting ? r_data[sl] : 1.0;
            const double ddl = acting ? r_ddl[sl] : 0.0;
            if (acting) atomicOr(s_mask, 1ull << rank);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const unsigned long long rmask = *s_mask;
            // acting slot = number of acting agents that arrived earlier; k = acting agents of the environment
            const int slot = acting ? __popcll(rmask & ((1ull << rank) - 1ull)) : -1;
            const int k = valid ? __popcll(rmask) : 0;

            if (p.explore && acting) {   // _maybe_apply_env_exploration, MRRL.PY:258-274
                if constexpr (TAPE) ex_u = tape.explore_u[env * n + slot];
                draws_lane += 1;
                if (!(ex_u > eps_k)) {
                    if constexpr (TAPE) {
                        ex_tgt = tape.explore_tgt[env * n + slot];
                        ex_res = tape.explore_res[env * n + slot];
                        if (ex_tgt < 0 || ex_tgt >= n) { atomicOr(err_flag, 1); ex_tgt = 0; }
                    }
                    tgt = ex_tgt;
                    res = (float)ex_res;
                    draws_lane += 2;
                }
            }

            double t_trans = 0.0;
            if (acting && tgt != sl) {   // MRRL.PY:409-414
                if constexpr (TAPE) { net_bw = tape.bw_u[env * n + slot]; net_lat = tape.net_lat[env * n + slot]; }
                draws_lane += 2;
                const double bw = net_bw * 1e6 / 8;
                t_trans = (data / bw) + net_lat;
            }
            // capacity of the target server: held by lane `tgt` of this group
            const int tl = gbase + (acting ? tgt : sl);
            const double cap_t = __shfl(cap_me, tl), icap_t = __shfl(icap_me, tl);
            const double alloc = cap_t * (double)res;        // MRRL.PY:418 (float64 * float32)
            const double gcyc = qdiv(alloc, 1e9, 1e-9);
            const double t_exec = wl / gcyc;                  // MRRL.PY:419
            const double energy = wl * gcyc;                  // MRRL.PY:423
            const double inc = t_exec * gcyc;                 // MRRL.PY:437

            STAMP(2);
            // ---- ordered chain over acting slots: queueing delay sees earlier placements (MRRL.PY:416,437)
            bool viol = false;
            for (int s = 0; __ballot(acting && slot >= s) != 0ull; ++s) {
                if (acting && slot == s) {
                    const double l = r_load[tgt];
                    const double t_queue = qdiv(l, cap_t, icap_t);
                    const double t_total = t_trans + t_queue + t_exec;
                    viol = (t_now + t_total) > ddl;          // MRRL.PY:431
                    s_lat[s] = t_total;
                    s_en[s] = energy;
                    r_load[tgt] = l + inc;
                    r_queue[tgt] += 1;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const int n_viol = __popcll(group_bits<LPE>(__ballot(viol), g, GW));

            STAMP(3);
            // ---- reward (MRRL.PY:326-338) and episode accumulators (MRRL.PY:463-466) ----------------
            const int tasks_before = r_int[I_TASKS];
            double sum_lat, sum_en, acc_lat = r_scal[S_SUMLAT], acc_en = r_scal[S_SUMEN];
            if (__ballot(k > 4) == 0ull) {
                // common case: one batch of LDS reads, sums in numpy's order for fewer than 8 addends
                const double a0 = s_lat[0], a1 = s_lat[1], a2 = s_lat[2], a3 = s_lat[3];
                const double e0 = s_en[0], e1 = s_en[1], e2 = s_en[2], e3 = s_en[3];
                double rl = -0.0, re = -0.0;
                if (k > 0) { rl += a0; re += e0; acc_lat += a0; acc_en += e0; }
                if (k > 1) { rl += a1; re += e1; acc_lat += a1; acc_en += e1; }
                if (k > 2) { rl += a2; re += e2; acc_lat += a2; acc_en += e2; }
                if (k > 3) { rl += a3; re += e3; acc_lat += a3; acc_en += e3; }
                sum_lat = 0.0 + rl;
                sum_en = 0.0 + re;
            } else {
                sum_lat = np_sum_small(s_lat, k);
                sum_en = np_sum_small(s_en, k);
                for (int i = 0; i < k; ++i) { acc_lat += s_lat[i]; acc_en += s_en[i]; }
            }
            if (k > 0) {
                const double kd = (double)k, ik = __shfl(ik_lane, gbase + k - 1);
                const double cost = (cc->w_lat * qdiv(sum_lat, kd, ik)) + (cc->w_en * qdiv(sum_en, kd, ik)) +
                                    (cc->w_viol * (double)n_viol);
                reward = -cost;
            }
            if (agent) {
                if (out.step_latency) out.step_latency[env * n + sl] = sl < k ? s_lat[sl] : __builtin_nan("");
                if (lat_log && sl < k) {
                    const int64_t at = (int64_t)tasks_before + sl;
                    if (at < p.lat_cap) lat_log[env * p.lat_cap + at] = s_lat[sl];
                }
            }
            if (valid && sl == 0) {
                r_scal[S_SUMLAT] = acc_lat;
                r_scal[S_SUMEN] = acc_en;
                r_int[I_TASKS] = tasks_before + k;
                r_int[I_VIOL] += n_viol;
                r_int[I_DONE] += k - n_viol;
                if (lat_log) {
                    const int64_t end = (int64_t)tasks_before + k;
                    r_int[I_LOGGED] = (int32_t)(end < p.lat_cap ? end : p.lat_cap);
                }
            }
        }
        STAMP(4);
    }

    __syncthreads();   // the scalar wavefront's windows and arrival counts of this step are in LDS
    if (scalar_wave) {
        if (ks + 1 < K) scalar_step(step_no + 1u, (ks + 1) & 1);    // one step ahead of the workers
        continue;
    }

    // =============================== worker wavefront, part 2: arrivals (MRRL.PY:340-383) ===============
    STAMP(5);
    const double window = s_win[(ks & 1) * 64 + wave * EPW + g];
    const int n_new = valid ? s_nnew[(ks & 1) * 64 + wave * EPW + g] : 0;
    float ratio_me = 0.0f;
    if (agent) {
        r_rank[sl] = -1;
        ratio_me = (float)qdiv(r_load[sl], cap_me, icap_me);   // load/cap BEFORE this step's decay
        s_ratio[sl] = ratio_me;
    }
    // arrivals 0..PAR-1 in parallel, arrival j on lane j of the group
    const int PAR = TAPE ? (GW < 16 ? GW : 16) : par_philox;
    const bool has = valid && sl < PAR && sl < n_new;
    double a_wl = pre_wl, a_data = pre_data, a_offs = pre_offs;
    int a_dec = pre_dec;
    if constexpr (TAPE) {
        if (has) {
            const int64_t o = env * tape.arrival_slots + sl;
            a_wl = tape.task_workload[o]; a_data = tape.task_data[o];
            a_offs = tape.task_deadline_offset[o]; a_dec = tape.task_decider[o];
            if (a_dec < 0 || a_dec >= n) { atomicOr(err_flag, 4); a_dec = 0; }
        }
    }
    const double a_deadline = t_now + a_offs;              // MRRL.PY:217
    if (has) atomicMin(&s_first[a_dec], sl);               // first arrival of each decider (MRRL.PY:358-362)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool is_new = has && s_first[a_dec] == sl;
    const uint32_t new_bits = (uint32_t)group_bits<LPE>(__ballot(is_new), g, GW);
    const int first_me = agent ? s_first[sl] : 0x7fffffff;
    const int q_me = agent ? r_queue[sl] : 0;              // queue length BEFORE this call's bumps (MRRL.PY:311)
    const bool pend_me = first_me != 0x7fffffff;
    if (is_new) {                                          // MRRL.PY:361 and the task columns of _get_obs
        r_rank[a_dec] = (int8_t)__popc(new_bits & ((1u << sl) - 1u));
        r_wl[a_dec] = a_wl; r_data[a_dec] = a_data; r_ddl[a_dec] = a_deadline;
        const double rem = a_deadline - t_now;
        float* row = s_obs + a_dec * w;
        row[2] = (float)a_data;
        row[3] = (float)a_wl;
        row[4] = (float)(rem > 0 ? rem : 0.0);
    }
    if (pend_me) { s_obs[sl * w] = ratio_me; s_obs[sl * w + 1] = (float)q_me; }
    // the other servers' load ratios: every lane writes its own ratio into the rows of the pending agents
    {
        uint64_t pm = group_bits<LPE>(__ballot(pend_me), g, GW);
        while (__ballot(pm != 0ull) != 0ull) {
            if (pm != 0ull) {
                const int r = __ffsll((unsigned long long)pm) - 1;
                pm &= pm - 1ull;
                if (agent && sl != r) s_obs[r * w + (sl < r ? sl + 5 : sl + 4)] = ratio_me;
            }
        }
    }
    if (has && !is_new) atomicAdd(&r_queue[a_dec], 1);    // MRRL.PY:359
    int n_pending = __popc(new_bits);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    STAMP(6);
    if (__ballot(n_new > PAR) != 0ull) {
        // rare tail: arrivals beyond the parallel batch, one at a time in arrival order
        for (int j = PAR;; ++j) {
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
                } else {
                    const uint4 a = philox(envid, p.episode, step_no, (P_TASK_A << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                    const uint4 b = philox(envid, p.episode, step_no, (P_TASK_B << 24) | (uint32_t)j, p.seed_lo, p.seed_hi);
                    wl = cc->wl_lo + cc->wl_span * u53(a.x, a.y);
                    data = cc->data_lo + cc->data_span * u53(a.z, a.w);
                    offs = cc->ddl_lo + cc->ddl_span * u53(b.x, b.y);
                    dec = (int)__umulhi(b.z, (uint32_t)n);
                }
                const double deadline = t_now + offs;          // MRRL.PY:217
                const bool fresh = r_rank[dec] < 0;            // broadcast read, before any lane updates
                const float q_now = (float)r_queue[dec];
                if (sl == 0) {
                    if (fresh) {                                // MRRL.PY:361
                        r_rank[dec] = (int8_t)n_pending;
                        r_wl[dec] = wl; r_data[dec] = data; r_ddl[dec] = deadline;
                    } else r_queue[dec] += 1;                  // MRRL.PY:359
                }
                if (fresh) {                                    // _get_obs, MRRL.PY:307-324
                    const double rem = deadline - t_now;
                    for (int c = sl; c < w; c += GW) {
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
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
    const int rank_out = agent ? (int)r_rank[sl] : -1;
    const uint64_t act_bits = group_bits<LPE>(__ballot(rank_out >= 0), g, GW);
    if (out.n_draws) {
        int d = draws_lane;
        if constexpr (LPE == 0) {
            const int mine = d;
            d = 0;
            for (int i = 0; i < GW; ++i) d += __shfl(mine, gbase + i);
        } else {
            for (int m = LPE / 2; m > 0; m >>= 1) d += __shfl_xor(d, m, LPE);
        }
        // once per environment: rate (random mode), window (step only), arrival count, 4 per arrival
        const int draws_env = (p.rate_mode == 2 ? 1 : 0) + (RESET ? 0 : 1) + 1 + 4 * n_new;
        if (valid && sl == 0) out.n_draws[env] = d + draws_env;
    }
    if (valid && sl == 0) {
        if constexpr (!RESET) {
            r_scal[S_TIME] = t_now + window;               // delta_t > 0 always: dt_min > 0
            r_scal[S_ELAPSED] = r_scal[S_ELAPSED] + window;
            r_int[I_STEP] = (int32_t)step_no;
            const int64_t he = env + (int64_t)ks * hist_stride;
            if (out.reward) out.reward[he] = reward;
            if (out.emitted) out.emitted[he] = n_pending > 0;
            if (out.reward_seen) out.reward_seen[he] = n_pending > 0 ? reward : 0.0;
            if (out.done) out.done[he] = (int32_t)step_no >= p.horizon;
        }
        r_int[I_PENDING] = n_pending;
        out.active[env + (int64_t)ks * obs_hist_stride] = act_bits;
    }
    if (agent && out.rank) out.rank[(env + (int64_t)ks * obs_hist_stride) * n + sl] = (int8_t)rank_out;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    STAMP(10);
    // ---- stage out: records and observation tile, 16-byte lane-linear stores -------------------------
    if (ks == K - 1) {      // the records leave LDS once per launch
        const uint4* src = reinterpret_cast<const uint4*>(wbase);
        uint4* dst = reinterpret_cast<uint4*>(state + env0 * p.rec_bytes);
        for (int i = lane; i < n_valid * p.rec_bytes / 16; i += 64) dst[i] = src[i];
    }
    if (ks == K - 1 || obs_hist_stride != 0) {
        const int64_t fbase = (env0 + (int64_t)ks * obs_hist_stride) * p.obs_floats;
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
    }   // steps of the launch
}

// ------------------------------------------------------------------------------------------------
// episode metrics (MRRL.PY:470-504); p99 from the latency log when present
// ------------------------------------------------------------------------------------------------
// the five metrics that need no sort, and the p99 handed in (MRRL.PY:470-504); one thread per environment calls it
__device__ inline void write_metrics(const EnvParams& p, const EnvConst* __restrict__ cc, const double* scal, const int32_t* ri,
                                     double p99, double* __restrict__ o)
{
    const int n = p.n;
    const int tasks = ri[I_TASKS];
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

// Episode metrics (MRRL.PY:470-504), one WAVEFRONT per environment.  The p99 needs sorted[lo] and sorted[hi] with
// lo = floor(0.99 (L - 1)), about 1 % of the latency log from the top: the log is held in registers and maxima are extracted
// one at a time (em_env_math.h: wave_p99; a dozen rounds for the default episode's ~830 tasks); the first version sorted
// the whole log per workgroup (a bitonic network of 55 barriers: 3.1 ms per episode at 65,536 environments, now 0.34).
constexpr int SEL_PER_LANE = 32;                 // logs up to 2,048 tasks in registers; larger ones stream from memory
__global__ __launch_bounds__(256) void metrics_select_kernel(const EnvParams p, const unsigned char* __restrict__ state,
                                                             const EnvConst* __restrict__ cc,
                                                             const double* __restrict__ lat_log, double* __restrict__ out6)
{
    const int lane = threadIdx.x & 63;
    const int64_t env = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (env >= p.n_envs) return;
    const int n = p.n;
    const unsigned char* rec = state + env * p.rec_bytes;
    const double* scal = reinterpret_cast<const double*>(rec + rec_off_scal(n));
    const int32_t* ri = reinterpret_cast<const int32_t*>(rec + rec_off_int(n));
    const int tasks = ri[I_TASKS];
    const int logged = lat_log ? ri[I_LOGGED] : 0;
    double p99 = __builtin_nan("");
    if (tasks == 0) p99 = 0.0;
    else if (lat_log && logged == tasks) {
        p99 = wave_p99_any<SEL_PER_LANE>(lat_log + env * p.lat_cap, logged, lane);
    }
    if (lane == 0) write_metrics(p, cc, scal, ri, p99, out6 + env * 6);
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
    int wg_waves = 8;              // wavefronts per workgroup: wg_waves - 1 workers + 1 scalar
    int32_t step = 0;              // steps taken since the last reset
    unsigned char* d_state = nullptr;
    EnvConst* d_const = nullptr;
    double* d_lat_log = nullptr;
    int32_t* d_err = nullptr;
    bool was_reset = false;
};

namespace {

int pow2_lanes(int n) { return n <= 16 ? 16 : (n <= 32 ? 32 : 64); }

// Automatic layout: packed groups of n lanes when that fits more environments into a wavefront and the batch is
// large enough to be throughput bound; power-of-two groups (shuffle butterflies, 8 parallel arrivals) otherwise.
int pick_lanes(int n, int64_t n_envs)
{
    const int p2 = pow2_lanes(n);
    if (n < 64 && 64 / n > 64 / p2 && n_envs >= EM_PACKED_MIN_ENVS) return EM_LANES_PACKED;
    return p2;
}

// Wavefronts per workgroup (workers + 1 scalar).  Small batches are bound by the critical path of ONE workgroup, where
// the workers' barrier wait grows with their number: 4 workers + 1 scalar measured best up to 16,384 environments
// (6.98 vs 7.43 us per launch at 4,096), 7 + 1 amortises the scalar wavefront best from 32,768 (16.4 vs 18.8 us there with the packed layout).
int pick_waves(int64_t n_envs) { return n_envs < EM_PACKED_MIN_ENVS ? 5 : 8; }

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
    const int epw = 64 / (lanes == EM_LANES_PACKED ? p.n : lanes);
    p.wave_bytes = wave_bytes_of(p.n, epw);
#ifdef EM_STAMP
    p.pad0 = em_stamp_cut;
#endif
    // _get_current_epsilon, MRRL.PY:249-256 (the several-steps-per-launch kernel re-evaluates it per step)
    double eps = 0.0;
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

template <bool TAPE, bool RESET, int ACT = ACT_DICT>
int launch_env(em_handle* h, const int32_t* placement, const float* resource, const em_tape* tape,
               const em_step_out* out, hipStream_t s, float greedy_share = 0.5f, int k_steps = 1, int history = 0)
{
    const int lanes = h->lanes;
    EnvParams p = make_params(h, lanes);
    p.greedy_share = greedy_share;
    MultiParams mp;
    std::memset(&mp, 0, sizeof mp);
    mp.k_steps = k_steps;
    mp.hist_stride = (history & EM_HISTORY_REWARDS) ? h->n_envs : 0;
    mp.obs_hist_stride = (history & EM_HISTORY_OBS) ? h->n_envs : 0;
    mp.explore_counter0 = h->explore_counter;
    const int epw = 64 / (lanes == EM_LANES_PACKED ? p.n : lanes);
    // the scalar wavefront serves one environment per lane: at most 64 environments per workgroup
    const int n_work = (h->wg_waves - 1) * epw <= 64 ? h->wg_waves - 1 : (64 / epw > 0 ? 64 / epw : 1);
    const int64_t epb = (int64_t)n_work * epw;           // environments per workgroup
    const int64_t blocks = (h->n_envs + epb - 1) / epb;
    // + the scalar wavefront's window/arrival-count buffers (two-deep in the several-steps kernel)
    const size_t lds = (size_t)p.wave_bytes * n_work + (k_steps > 1 ? 2 : 1) * 64 * 12 + 32;   // + chunk totals, arrival count
    if (lds > 160 * 1024) return em::fail(EM_ERR_INVALID, "n_nodes too large for the LDS tile (%zu bytes)", lds);
    em_tape tp;
    std::memset(&tp, 0, sizeof tp);
    if (tape) tp = *tape;
    const dim3 grid((unsigned)blocks), block(64 * (n_work + 1));
#define EM_LAUNCH1(L, NC)                                                                                              \
    do {                                                                                                               \
        if (lds > 64 * 1024) {                                                                                         \
            const hipError_t e = hipFuncSetAttribute((const void*)env_kernel<L, TAPE, RESET, ACT, NC>,                 \
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
            if (e != hipSuccess) return em::hip_fail(e, "hipFuncSetAttribute");                                        \
        }                                                                                                              \
        hipLaunchKernelGGL((env_kernel<L, TAPE, RESET, ACT, NC>), grid, block, lds, s, p, h->d_state, h->d_const,      \
                           placement, resource, tp, *out, h->d_lat_log, h->d_err);                                     \
    } while (0)
#define EM_LAUNCH(L)                                                                                                   \
    do {                                                                                                               \
        if (k_steps > 1) {                                                                                             \
            if constexpr (!TAPE && !RESET) {                                                                           \
                if (lds > 64 * 1024) {                                                                                 \
                    const hipError_t e = hipFuncSetAttribute((const void*)env_multi_kernel<L, ACT>,                   \
                                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
                    if (e != hipSuccess) return em::hip_fail(e, "hipFuncSetAttribute");                                \
                }                                                                                                      \
                hipLaunchKernelGGL((env_multi_kernel<L, ACT>), grid, block, lds, s, p, mp, h->d_state, h->d_const,    \
                                   placement, resource, tp, *out, h->d_lat_log, h->d_err);                             \
            }                                                                                                          \
        } else {                                                                                                       \
            bool fixed_n = false;                                                                                      \
            if constexpr (L == 0 || L == 16) {                                                                         \
                if (p.n == 10) {                                                                                       \
                    fixed_n = true;                                                                                    \
                    EM_LAUNCH1(L, 10);                                                                                 \
                }                                                                                                      \
            }                                                                                                          \
            if (!fixed_n) EM_LAUNCH1(L, 0);                                                                            \
        }                                                                                                              \
    } while (0)
    if (k_steps > 1 && (TAPE || RESET)) return em::fail(EM_ERR_INVALID, "several steps per launch need the in-kernel random stream");
    if (lanes == EM_LANES_PACKED) EM_LAUNCH(0);
    else if (lanes == 16) EM_LAUNCH(16);
    else if (lanes == 32) EM_LAUNCH(32);
    else EM_LAUNCH(64);
#undef EM_LAUNCH
#undef EM_LAUNCH1
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

namespace {
template <int ACT>
int step_with_policy(em_handle* h, const float* raw, float share, const em_tape* tape, const em_step_out* out, void* stream,
                     const char* what)
{
    if (!h) return em::fail(EM_ERR_INVALID, "%s: null handle", what);
    if (!h->was_reset) return em::fail(EM_ERR_STATE, "%s before em_reset", what);
    int rc = check_out(h, out, true);
    if (rc) return rc;
    rc = check_tape(h, tape, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    rc = tape ? launch_env<true, false, ACT>(h, nullptr, raw, tape, out, s, share)
              : launch_env<false, false, ACT>(h, nullptr, raw, nullptr, out, s, share);
    if (rc) return rc;
    h->step += 1;
    if (!h->explore_off) h->explore_counter += 1;   // MRRL.PY:467-468
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
    h->lanes = pick_lanes(cfg->n_nodes, n_envs);
    h->wg_waves = pick_waves(n_envs);
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
    h->step = 0;
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
    h->step += 1;
    if (!h->explore_off) h->explore_counter += 1;   // MRRL.PY:467-468
    return EM_OK;
}

int em_step_many(em_handle* h, int32_t k_steps, int32_t action_mode, const int32_t* placement, const float* resource_or_raw,
                 float greedy_share, int32_t history, const em_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "em_step_many: null handle");
    if (!h->was_reset) return em::fail(EM_ERR_STATE, "em_step_many before em_reset");
    if (k_steps < 1 || (int64_t)h->step + k_steps > h->cfg.episode_length)
        return em::fail(EM_ERR_INVALID, "em_step_many: %d steps from step %d leave the episode of %d steps", k_steps, h->step,
                        h->cfg.episode_length);
    if (history & ~(EM_HISTORY_REWARDS | EM_HISTORY_OBS)) return em::fail(EM_ERR_INVALID, "em_step_many: unknown history bits");
    int rc = check_out(h, out, true);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    switch (action_mode) {
    case EM_ACTIONS_DICT:
        if (!placement || !resource_or_raw) return em::fail(EM_ERR_INVALID, "em_step_many: placement/resource are required");
        rc = launch_env<false, false, ACT_DICT>(h, placement, resource_or_raw, nullptr, out, s, 0.5f, k_steps, history);
        break;
    case EM_ACTIONS_RAW:
        if (!resource_or_raw) return em::fail(EM_ERR_INVALID, "em_step_many: raw actions are required");
        rc = launch_env<false, false, ACT_RAW>(h, nullptr, resource_or_raw, nullptr, out, s, 0.5f, k_steps, history);
        break;
    case EM_ACTIONS_GREEDY:
        if (!(greedy_share > 0.0f) || !(greedy_share <= 1.0f))
            return em::fail(EM_ERR_INVALID, "em_step_many: resource share must be in (0, 1]");
        rc = launch_env<false, false, ACT_GREEDY>(h, nullptr, nullptr, nullptr, out, s, greedy_share, k_steps, history);
        break;
    default: return em::fail(EM_ERR_INVALID, "em_step_many: unknown action mode %d", action_mode);
    }
    if (rc) return rc;
    h->step += k_steps;
    if (!h->explore_off) h->explore_counter += k_steps;   // MRRL.PY:467-468, once per step
    return EM_OK;
}

int em_step_raw(em_handle* h, const float* raw_actions, const em_tape* tape, const em_step_out* out, void* stream)
{
    if (!raw_actions) return em::fail(EM_ERR_INVALID, "em_step_raw: raw_actions is required");
    return step_with_policy<ACT_RAW>(h, raw_actions, 0.5f, tape, out, stream, "em_step_raw");
}

int em_step_greedy(em_handle* h, float resource_share, const em_tape* tape, const em_step_out* out, void* stream)
{
    if (!(resource_share > 0.0f) || !(resource_share <= 1.0f))
        return em::fail(EM_ERR_INVALID, "em_step_greedy: resource_share must be in (0, 1]");
    return step_with_policy<ACT_GREEDY>(h, nullptr, resource_share, tape, out, stream, "em_step_greedy");
}

int em_episode_metrics(em_handle* h, double* out6, void* stream)
{
    if (!h || !out6) return em::fail(EM_ERR_INVALID, "em_episode_metrics: null argument");
    const EnvParams p = make_params(h, h->lanes);
    hipLaunchKernelGGL(metrics_select_kernel, dim3((unsigned)((h->n_envs + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, h->d_state,
                       h->d_const, h->d_lat_log, out6);
    return em::hip_check(hipGetLastError(), "metrics_select_kernel launch");
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
    if (lanes == 0) { h->lanes = pick_lanes(h->cfg.n_nodes, h->n_envs); return EM_OK; }
    if (lanes == EM_LANES_PACKED) {
        if (h->cfg.n_nodes >= 64) return em::fail(EM_ERR_INVALID, "the packed layout needs n_nodes < 64");
        h->lanes = lanes;
        return EM_OK;
    }
    if ((lanes != 16 && lanes != 32 && lanes != 64) || lanes < h->cfg.n_nodes)
        return em::fail(EM_ERR_INVALID, "lanes per env must be 1 (packed), 16, 32 or 64 and >= n_nodes");
    h->lanes = lanes;
    return EM_OK;
}

int em_get_kernel_variant(const em_handle* h, int32_t* lanes_per_env, int32_t* workgroup_waves)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    const int n = h->cfg.n_nodes;
    const int epw = 64 / (h->lanes == EM_LANES_PACKED ? n : h->lanes);
    // the launch clamps the workers so that the scalar wavefront's 64 lanes cover the workgroup's environments
    const int n_work = (h->wg_waves - 1) * epw <= 64 ? h->wg_waves - 1 : (64 / epw > 0 ? 64 / epw : 1);
    if (lanes_per_env) *lanes_per_env = h->lanes == EM_LANES_PACKED ? n : h->lanes;
    if (workgroup_waves) *workgroup_waves = n_work + 1;
    return h->lanes == EM_LANES_PACKED ? 1 : 0;
}

int em_set_workgroup_waves(em_handle* h, int waves)
{
    if (!h) return em::fail(EM_ERR_INVALID, "null handle");
    if (waves == 0) waves = pick_waves(h->n_envs);
    if (waves < 2 || waves * 64 > EM_ENV_MAX_THREADS) return em::fail(EM_ERR_INVALID, "workgroup wavefronts must be 2..%d", EM_ENV_MAX_THREADS / 64);
    h->wg_waves = waves;
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
