// edge_marl_env.hip - EdgeMARLEnv (the PyMARL twin environment, pymarl/src/envs/edge_marl_env.py), batched.
//
// One LANE per environment: the placements of a step form an ordered chain over the agents (each reads the load its
// predecessors left on the target, :106,127), n is small (10 in config/envs/edge_marl.yaml), and every agent acts every
// step, so there is no slot compaction to parallelise inside an environment.  The per-environment state lives in HBM as
// structure-of-arrays [field][agent][env]: a wavefront's loads and stores are 512-byte coalesced rows.  During the
// chain the loads and queue lengths sit in LDS ([agent][lane], conflict-free for any target), because the target index
// is data dependent.  The observation tensor ([env][agent][n+4] f32, which read flat is also get_state, :173-174) is
// staged through the same LDS and written by the whole workgroup as one contiguous range.
//
// float64 throughout, -ffp-contract=off, the reference's operation order (oracle/edge_marl_oracle.py restates it and is
// pinned to recorded reference episodes); np.mean / np.sum follow numpy's pairwise order for n <= 128.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <new>

#include "edgemarl.h"
#include "em_common.h"
#include "em_env_math.h"

#pragma clang fp contract(off)

namespace {

struct MarlConst {
    double cap[EM_MAX_NODES];
    double rate_fixed, rate_lo, rate_hi, cycle;
    double wl_lo, wl_hi, data_lo, data_hi, ddl_lo, ddl_hi, bw_lo, bw_hi, lat_lo, lat_hi;
    double alpha, beta, gamma, dt_min, dt_max;
    int32_t n, limit, arrival_mode, action_mode;
};

struct MarlState {          // HBM, [agent][env] unless noted
    double* load;
    float* queue;           // integer valued (counts), exact in f32
    double* wl;
    double* data;
    double* deadline;
    double* sim_time;       // [env]
    int32_t* step;          // [env]
};

struct MarlArgs {
    MarlState st;
    const MarlConst* kc;
    const int32_t* act_i;
    const float* act_f;
    emm_tape tape;
    emm_step_out out;
    int64_t n_envs;
    uint64_t env_offset;
    uint32_t key0, key1, episode;
    int32_t has_tape;
};

// numpy's add.reduce order for n <= 128, fed one element at a time (np.mean of edge_marl_env.py:138-140, np.sum of :298-299)
struct NpAcc {
    double r[8];
    double res;
    __device__ inline void add(int i, int n, double v)
    {
        const int nblk = n < 8 ? 0 : n - (n & 7);
        if (i < nblk) {
            const int sel = i & 7;
            const bool first = i < 8;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (sel == k) r[k] = first ? v : r[k] + v;
        } else {
            if (i == nblk) res = nblk ? ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7])) : 0.0;
            res += v;
        }
    }
    __device__ inline double total(int n) const
    {
        if (n >= 8 && (n & 7) == 0) return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        return n == 0 ? 0.0 : res;
    }
};

__device__ inline double uni(double u, double lo, double hi) { return lo + (hi - lo) * u; }

// LDS, rows T+1 apart: [n][T] float64 loads (reused as {ratio f32, deadline-remaining f32}) | [n][T] f32 queue | [n][T] f32 data | [n][T] f32 workload
template <bool RESET>
__global__ void marl_env_kernel(const MarlArgs a)
{
    extern __shared__ double lds_raw[];
    const int T = blockDim.x, TS = T + 1, tid = threadIdx.x;
    const MarlConst* kc = a.kc;
    const int n = kc->n;
    double* l_load = lds_raw;
    float* l_queue = reinterpret_cast<float*>(l_load + (size_t)n * TS);
    float* l_data = l_queue + (size_t)n * TS;
    float* l_wl = l_data + (size_t)n * TS;
    const int64_t B = a.n_envs, b0 = (int64_t)blockIdx.x * T, b = b0 + tid;
    const bool live = b < B;
    const uint32_t env_id_lo = (uint32_t)(a.env_offset + (uint64_t)b);

    double sim_time = 0.0;
    int step = 0;
    if (live) {
        if constexpr (!RESET) {
            sim_time = a.st.sim_time[b];
            step = a.st.step[b] + 1;
            for (int j = 0; j < n; ++j) {
                l_load[j * TS + tid] = a.st.load[(int64_t)j * B + b];
                l_queue[j * TS + tid] = a.st.queue[(int64_t)j * B + b];
            }
            // _sample_time_window (:230-234) with _current_task_rate (:236-243)
            double raw;
            if (a.has_tape) {
                raw = a.tape.dt_raw[b];
            } else {
                const uint4 blk = philox(env_id_lo, a.episode, (uint32_t)step, (uint32_t)P_DT << 24, a.key0, a.key1);
                double rate;
                if (kc->arrival_mode == 1) {
                    const double phase = fmod(sim_time, kc->cycle) / kc->cycle;
                    rate = kc->rate_lo + (kc->rate_hi - kc->rate_lo) * 0.5 * (1 + det_sin2pi(phase - 0.25));
                } else if (kc->arrival_mode == 2) {
                    rate = uni(u53(blk.z, blk.w), kc->rate_lo, kc->rate_hi);
                } else {
                    rate = kc->rate_fixed;
                }
                const double scale = 1.0 / (rate > 1e-6 ? rate : 1e-6);
                raw = scale * (-det_log(1.0 - u53(blk.x, blk.y)));
            }
            const double window = raw < kc->dt_min ? kc->dt_min : (raw > kc->dt_max ? kc->dt_max : raw);
            sim_time = sim_time + window;

            NpAcc s_lat, s_en;
            int n_viol = 0;
            // the next agent's task is fetched while the current one is evaluated (one lane cannot hide HBM latency otherwise)
            const double* __restrict__ g_wl = a.st.wl;
            const double* __restrict__ g_data = a.st.data;
            const double* __restrict__ g_ddl = a.st.deadline;
            double wl_nx = g_wl[b], data_nx = g_data[b], ddl_nx = g_ddl[b];
            for (int i = 0; i < n; ++i) {
                const double wl = wl_nx, data = data_nx, deadline = ddl_nx;
                if (i + 1 < n) {
                    wl_nx = g_wl[(int64_t)(i + 1) * B + b];
                    data_nx = g_data[(int64_t)(i + 1) * B + b];
                    ddl_nx = g_ddl[(int64_t)(i + 1) * B + b];
                }
                int target;
                double share;
                if (kc->action_mode == 0) {
                    const int raw_a = a.act_i[b * n + i];
                    target = raw_a < 0 ? 0 : (raw_a > n - 1 ? n - 1 : raw_a);                 // :101
                    share = 0.5;
                } else {
                    const float* v = a.act_f + (b * n + i) * (int64_t)(n + 1);              // :304-318
                    bool all_zero = true, have_nan = false;
                    float best = v[0];
                    int bi = 0;
                    for (int j = 0; j < n; ++j) {
                        const float x = v[j];
                        all_zero = all_zero && (fabs((double)x) <= 1e-8);
                        if (!have_nan && (x != x)) { have_nan = true; bi = j; }
                        if (!have_nan && x > best) { best = x; bi = j; }
                    }
                    target = all_zero ? 0 : bi % n;
                    double nrm = ((double)v[n] + 1.0) / 2.0;
                    nrm = nrm < 0.0 ? 0.0 : (nrm > 1.0 ? 1.0 : nrm);
                    share = 0.1 + 0.9 * nrm;
                }
                const double placement_load = l_load[target * TS + tid];
                double transfer = 0.0;
                if (target != i) {
                    double bw_u, net_lat;
                    if (a.has_tape) {
                        bw_u = a.tape.bw_u[b * n + i];
                        net_lat = a.tape.net_lat[b * n + i];
                    } else {
                        const uint4 blk = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_NET << 24) | (uint32_t)i, a.key0, a.key1);
                        bw_u = uni(u53(blk.x, blk.y), kc->bw_lo, kc->bw_hi);
                        net_lat = uni(u53(blk.z, blk.w), kc->lat_lo, kc->lat_hi);
                    }
                    const double bw = bw_u * 1e6 / 8.0;
                    transfer = (data / bw) + net_lat;                                       // :111-113
                }
                const double cap_t = kc->cap[target];
                const double queue_lat = placement_load / (cap_t + 1e-9);                   // :115
                const double eff = cap_t * share;
                const double eff_g = eff / 1e9;
                const double exec_lat = wl / eff_g;                                         // :117
                const double total = transfer + queue_lat + exec_lat;
                const double energy = wl * eff_g;                                           // :120
                s_lat.add(i, n, total);
                s_en.add(i, n, energy);
                n_viol += (sim_time + total) > deadline ? 1 : 0;                            // :125
                l_load[target * TS + tid] = placement_load + wl;                             // :127
                l_queue[target * TS + tid] += 1.0f;
                if (a.out.latency) a.out.latency[b * n + i] = total;
            }
            // _advance_system (:256-259) and the Jain index of load + 1e-6 (:143,296-302)
            NpAcc s_v, s_v2;
            for (int j = 0; j < n; ++j) {
                const double processed = kc->cap[j] * window;
                double l = l_load[j * TS + tid] - processed;
                l = l > 0.0 ? l : 0.0;
                float q = l_queue[j * TS + tid] - 1.0f;
                q = q > 0.0f ? q : 0.0f;
                l_load[j * TS + tid] = l;
                l_queue[j * TS + tid] = q;
                a.st.load[(int64_t)j * B + b] = l;
                a.st.queue[(int64_t)j * B + b] = q;
                const double v = l + 1e-6;
                s_v.add(j, n, v);
                s_v2.add(j, n, v * v);
            }
            const double dn = (double)n;
            const double avg_lat = s_lat.total(n) / dn, avg_en = s_en.total(n) / dn, viol_rate = (double)n_viol / dn;
            const double throughput = window > 0.0 ? dn / window : 0.0;                     // :142
            const double sv = s_v.total(n);
            const double den = dn * s_v2.total(n);
            const double jain = den <= 0.0 ? 0.0 : (sv * sv) / den;
            const double reward = -(kc->alpha * avg_lat + kc->beta * avg_en + kc->gamma * viol_rate);   // :290-294
            if (a.out.reward) a.out.reward[b] = reward;
            if (a.out.terminated) a.out.terminated[b] = step >= kc->limit ? 1 : 0;          // :146
            if (a.out.info) {
                double* o = a.out.info + b * 6;
                o[0] = avg_lat * 1000.0; o[1] = avg_en; o[2] = viol_rate; o[3] = throughput; o[4] = jain; o[5] = avg_lat;
            }
        } else {
            for (int j = 0; j < n; ++j) {                                                   // reset, :67-73
                l_load[j * TS + tid] = 0.0;
                l_queue[j * TS + tid] = 0.0f;
                a.st.load[(int64_t)j * B + b] = 0.0;
                a.st.queue[(int64_t)j * B + b] = 0.0f;
            }
        }
        a.st.sim_time[b] = sim_time;
        a.st.step[b] = step;

        // _spawn_new_tasks / _sample_task (:245-254,261-263) and the per-agent part of _build_obs (:265-276)
        for (int i = 0; i < n; ++i) {
            double wl, data, off;
            if (a.has_tape) {
                wl = a.tape.task_workload[b * n + i];
                data = a.tape.task_data[b * n + i];
                off = a.tape.task_deadline_offset[b * n + i];
            } else {
                const uint4 ba = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_TASK_A << 24) | (uint32_t)i, a.key0, a.key1);
                const uint4 bb = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_TASK_B << 24) | (uint32_t)i, a.key0, a.key1);
                wl = uni(u53(ba.x, ba.y), kc->wl_lo, kc->wl_hi);
                data = uni(u53(ba.z, ba.w), kc->data_lo, kc->data_hi);
                off = uni(u53(bb.x, bb.y), kc->ddl_lo, kc->ddl_hi);
            }
            const double deadline = sim_time + off;
            a.st.wl[(int64_t)i * B + b] = wl;
            a.st.data[(int64_t)i * B + b] = data;
            a.st.deadline[(int64_t)i * B + b] = deadline;
            const double rem = deadline - sim_time;
            const double ratio = l_load[i * TS + tid] / (kc->cap[i] + 1e-9);                 // :266
            float2 pr;
            pr.x = (float)ratio;
            pr.y = (float)(rem > 0.0 ? rem : 0.0);
            *reinterpret_cast<float2*>(&l_load[i * TS + tid]) = pr;
            l_data[i * TS + tid] = (float)data;
            l_wl[i * TS + tid] = (float)wl;
        }
    }
    if (!a.out.obs) return;
    __syncthreads();
    // the workgroup's observation rows are one contiguous range of the output, [env][agent][n+4]: a thread owns a fixed
    // (agent, column) and walks the environments, so consecutive lanes store consecutive floats and the index arithmetic
    // is done once (rows are TS = T+1 apart in LDS: lanes that read different agents of one environment hit different banks)
    const int w = n + 4, per_env = n * w;
    const int envs_here = (int)(B - b0 < T ? B - b0 : T);
    float* dst = a.out.obs + b0 * per_env;
    const float2* l_pair = reinterpret_cast<const float2*>(l_load);
    constexpr int SPLIT = 8;                            // environment ranges per (agent, column): work for every lane
    const int chunk = (envs_here + SPLIT - 1) / SPLIT;
    for (int item = tid; item < per_env * SPLIT; item += T) {
        const int part = item / per_env, base = item - part * per_env;
        const int i = base / w, c = base - i * w;
        const int el0 = part * chunk, el1 = el0 + chunk < envs_here ? el0 + chunk : envs_here;
        const float* src;
        int stride_f;                                   // LDS stride between environments, in floats
        if (c == 0) { src = &l_pair[i * TS].x; stride_f = 2; }
        else if (c == 1) { src = &l_queue[i * TS]; stride_f = 1; }
        else if (c == 2) { src = &l_data[i * TS]; stride_f = 1; }
        else if (c == 3) { src = &l_wl[i * TS]; stride_f = 1; }
        else if (c == 4) { src = &l_pair[i * TS].y; stride_f = 2; }
        else { const int j = c - 5; src = &l_pair[(j < i ? j : j + 1) * TS].x; stride_f = 2; }   // neighbours in index order, :278-282
        float* d = dst + base;
        for (int el = el0; el < el1; ++el) d[(int64_t)el * per_env] = src[el * stride_f];
    }
}


// ------------------------------------------------------------------------------------------------
// Small batches: 16 lanes per environment (n <= 16), lane = agent.  With a few thousand environments one lane per
// environment leaves most SIMDs idle and a step costs the serial latency of n Philox blocks and ~5n float64 divisions
// (35 us at 4,096 environments).  Here the draws, divisions and observation rows of the agents run side by side; only the
// load chain itself is serial (three lane shuffles per agent).  Same state arrays, same arithmetic, same results.
// ------------------------------------------------------------------------------------------------
constexpr int G16 = 16, ENVS16 = 16;      // lanes per environment, environments per 256-lane workgroup

template <bool RESET>
__global__ __launch_bounds__(256) void marl_env16_kernel(const MarlArgs a)
{
    __shared__ double s_val[ENVS16][4][G16];      // per agent: total latency, energy, load + 1e-6, its square
    __shared__ float s_obs[ENVS16][5][G16];       // per agent: load ratio, queue, data, workload, deadline remaining
    const int tid = threadIdx.x, lane = tid & 63, i = lane & (G16 - 1), gbase = lane & 48;
    const int el = (tid >> 6) * 4 + (lane >> 4);
    const MarlConst* kc = a.kc;
    const int n = kc->n;
    const int64_t B = a.n_envs, b0 = (int64_t)blockIdx.x * ENVS16, b = b0 + el;
    const bool env_ok = b < B, ok = env_ok && i < n;
    const int64_t bb = env_ok ? b : B - 1;        // idle lanes compute on a valid address and store nothing
    const int ic = i < n ? i : n - 1;
    const int64_t at = (int64_t)ic * B + bb;
    const uint32_t env_id_lo = (uint32_t)(a.env_offset + (uint64_t)bb);
    const double cap_i = kc->cap[ic];

    double sim_time = 0.0, load = 0.0;
    float queue = 0.0f;
    int step = 0;
    if constexpr (!RESET) {
        sim_time = a.st.sim_time[bb];
        step = a.st.step[bb] + 1;
        load = a.st.load[at];
        queue = a.st.queue[at];
        const double wl = a.st.wl[at], data = a.st.data[at], deadline = a.st.deadline[at];
        double raw;                                                                          // :230-243, every lane of the environment
        if (a.has_tape) {
            raw = a.tape.dt_raw[bb];
        } else {
            const uint4 blk = philox(env_id_lo, a.episode, (uint32_t)step, (uint32_t)P_DT << 24, a.key0, a.key1);
            double rate;
            if (kc->arrival_mode == 1) {
                const double phase = fmod(sim_time, kc->cycle) / kc->cycle;
                rate = kc->rate_lo + (kc->rate_hi - kc->rate_lo) * 0.5 * (1 + det_sin2pi(phase - 0.25));
            } else if (kc->arrival_mode == 2) {
                rate = uni(u53(blk.z, blk.w), kc->rate_lo, kc->rate_hi);
            } else {
                rate = kc->rate_fixed;
            }
            const double scale = 1.0 / (rate > 1e-6 ? rate : 1e-6);
            raw = scale * (-det_log(1.0 - u53(blk.x, blk.y)));
        }
        const double window = raw < kc->dt_min ? kc->dt_min : (raw > kc->dt_max ? kc->dt_max : raw);
        sim_time = sim_time + window;

        int target;
        double share;
        if (kc->action_mode == 0) {
            const int raw_a = a.act_i[bb * n + ic];
            target = raw_a < 0 ? 0 : (raw_a > n - 1 ? n - 1 : raw_a);                         // :101
            share = 0.5;
        } else {
            const float* v = a.act_f + (bb * n + ic) * (int64_t)(n + 1);                      // :304-318
            bool all_zero = true, have_nan = false;
            float best = v[0];
            int bi = 0;
            for (int j = 0; j < n; ++j) {
                const float x = v[j];
                all_zero = all_zero && (fabs((double)x) <= 1e-8);
                if (!have_nan && (x != x)) { have_nan = true; bi = j; }
                if (!have_nan && x > best) { best = x; bi = j; }
            }
            target = all_zero ? 0 : bi % n;
            double nrm = ((double)v[n] + 1.0) / 2.0;
            nrm = nrm < 0.0 ? 0.0 : (nrm > 1.0 ? 1.0 : nrm);
            share = 0.1 + 0.9 * nrm;
        }
        if (i >= n) target = 0;
        double bw_u, net_lat;
        if (a.has_tape) {
            bw_u = a.tape.bw_u[bb * n + ic];
            net_lat = a.tape.net_lat[bb * n + ic];
        } else {
            const uint4 blk = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_NET << 24) | (uint32_t)ic, a.key0, a.key1);
            bw_u = uni(u53(blk.x, blk.y), kc->bw_lo, kc->bw_hi);
            net_lat = uni(u53(blk.z, blk.w), kc->lat_lo, kc->lat_hi);
        }
        double transfer = 0.0;
        if (target != i) {
            const double bw = bw_u * 1e6 / 8.0;
            transfer = (data / bw) + net_lat;                                                 // :111-113
        }
        const double cap_t = kc->cap[target];
        const double eff = cap_t * share;
        const double eff_g = eff / 1e9;
        const double exec_lat = wl / eff_g;                                                   // :117
        const double energy = wl * eff_g;                                                     // :120
        // the ordered chain (:106,127): agent k reads the load its predecessors left on its target
        double my_pl = 0.0;
        for (int k = 0; k < n; ++k) {
            const int t_k = __shfl(target, gbase + k);
            const double w_k = __shfl(wl, gbase + k);
            const double pl_k = __shfl(load, gbase + t_k);
            if (i == k) my_pl = pl_k;
            if (i == t_k) { load = pl_k + w_k; queue += 1.0f; }
        }
        const double queue_lat = my_pl / (cap_t + 1e-9);                                      // :115
        const double total = transfer + queue_lat + exec_lat;
        const bool viol = ok && (sim_time + total) > deadline;                                // :125
        const int n_viol = __popcll((__ballot(viol) >> gbase) & 0xFFFFull);
        if (ok && a.out.latency) a.out.latency[b * n + i] = total;
        // _advance_system (:256-259)
        double l = load - cap_i * window;
        l = l > 0.0 ? l : 0.0;
        float q = queue - 1.0f;
        q = q > 0.0f ? q : 0.0f;
        load = l;
        queue = q;
        const double v = l + 1e-6;
        s_val[el][0][i] = total; s_val[el][1][i] = energy; s_val[el][2][i] = v; s_val[el][3][i] = v * v;
        __syncthreads();
        const double mine = np_sum_small(&s_val[el][i & 3][0], n);                            // lanes 0..3: the four ordered sums
        const double sum_lat = __shfl(mine, gbase), sum_en = __shfl(mine, gbase + 1);
        const double sv = __shfl(mine, gbase + 2), sv2 = __shfl(mine, gbase + 3);
        if (ok) {
            a.st.load[at] = l;
            a.st.queue[at] = q;
        }
        if (env_ok && i == 0) {
            const double dn = (double)n;
            const double avg_lat = sum_lat / dn, avg_en = sum_en / dn, viol_rate = (double)n_viol / dn;
            const double throughput = window > 0.0 ? dn / window : 0.0;                       // :142
            const double den = dn * sv2;
            const double jain = den <= 0.0 ? 0.0 : (sv * sv) / den;                           // :296-302
            const double reward = -(kc->alpha * avg_lat + kc->beta * avg_en + kc->gamma * viol_rate);
            if (a.out.reward) a.out.reward[b] = reward;
            if (a.out.terminated) a.out.terminated[b] = step >= kc->limit ? 1 : 0;
            if (a.out.info) {
                double* o = a.out.info + b * 6;
                o[0] = avg_lat * 1000.0; o[1] = avg_en; o[2] = viol_rate; o[3] = throughput; o[4] = jain; o[5] = avg_lat;
            }
        }
    } else {
        if (ok) {
            a.st.load[at] = 0.0;
            a.st.queue[at] = 0.0f;
        }
    }
    if (env_ok && i == 0) {
        a.st.sim_time[b] = sim_time;
        a.st.step[b] = step;
    }
    {   // _sample_task (:245-254) for this lane's agent, and its part of _build_obs (:265-276)
        double wl, data, off;
        if (a.has_tape) {
            wl = a.tape.task_workload[bb * n + ic];
            data = a.tape.task_data[bb * n + ic];
            off = a.tape.task_deadline_offset[bb * n + ic];
        } else {
            const uint4 ba = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_TASK_A << 24) | (uint32_t)ic, a.key0, a.key1);
            const uint4 bq = philox(env_id_lo, a.episode, (uint32_t)step, ((uint32_t)P_TASK_B << 24) | (uint32_t)ic, a.key0, a.key1);
            wl = uni(u53(ba.x, ba.y), kc->wl_lo, kc->wl_hi);
            data = uni(u53(ba.z, ba.w), kc->data_lo, kc->data_hi);
            off = uni(u53(bq.x, bq.y), kc->ddl_lo, kc->ddl_hi);
        }
        const double deadline = sim_time + off;
        if (ok) {
            a.st.wl[at] = wl;
            a.st.data[at] = data;
            a.st.deadline[at] = deadline;
        }
        const double rem = deadline - sim_time;
        s_obs[el][0][i] = (float)(load / (cap_i + 1e-9));                                     // :266
        s_obs[el][1][i] = queue;
        s_obs[el][2][i] = (float)data;
        s_obs[el][3][i] = (float)wl;
        s_obs[el][4][i] = (float)(rem > 0.0 ? rem : 0.0);
    }
    if (!a.out.obs) return;
    __syncthreads();
    const unsigned w = n + 4, per_env = n * w;
    const unsigned envs_here = (unsigned)(B - b0 < ENVS16 ? B - b0 : ENVS16);
    float* dst = a.out.obs + b0 * per_env;
    for (unsigned e = tid; e < envs_here * per_env; e += 256) {
        const unsigned eo = e / per_env, r = e - eo * per_env, ai = r / w, c = r - ai * w;
        float v;
        if (c < 5) v = s_obs[eo][c][ai];
        else { const unsigned j = c - 5; v = s_obs[eo][0][j < ai ? j : j + 1]; }              // neighbours in index order, :278-282
        dst[e] = v;
    }
}

}  // namespace

struct emm_handle {
    emm_config cfg;
    int64_t n_envs;
    int device;
    uint64_t seed, env_offset;
    uint32_t episode;
    MarlConst* kc_dev;
    void* state_block;
    MarlState st;
    int threads;
    size_t lds_bytes;
    int lanes_per_env;      // 1 or 16 (kernel variant; same state arrays)
};

namespace {

int check_cfg(const emm_config* c)
{
    if (c->n_nodes < 1 || c->n_nodes > EM_MAX_NODES) return em::fail(EM_ERR_INVALID, "emm_create: n_nodes %d outside [1,%d]", c->n_nodes, EM_MAX_NODES);
    if (c->episode_length < 1) return em::fail(EM_ERR_INVALID, "emm_create: episode_length must be positive");
    if (c->arrival_mode < 0 || c->arrival_mode > 2) return em::fail(EM_ERR_INVALID, "emm_create: arrival_mode must be 0, 1 or 2");
    if (c->action_mode < 0 || c->action_mode > 1) return em::fail(EM_ERR_INVALID, "emm_create: action_mode must be 0 (discrete) or 1 (hybrid)");
    if (c->arrival_mode == 1 && !(c->arrival_cycle_seconds > 0.0)) return em::fail(EM_ERR_INVALID, "emm_create: cyclic arrivals need a positive cycle");
    for (int i = 0; i < c->n_nodes; ++i)
        if (!(c->node_cpu_capacity[i] > 0.0)) return em::fail(EM_ERR_INVALID, "emm_create: node_cpu_capacity[%d] must be positive", i);
    return EM_OK;
}

// 16 lanes per environment up to 32,768 environments (8 vs 35 us at 4,096; 46 vs 41 us at 65,536) (latency bound there), one lane per environment beyond (issue bound)
int pick_lanes(int n, int64_t n_envs) { return (n <= G16 && n_envs <= 32768) ? 16 : 1; }

int launch(emm_handle* h, bool reset, const int32_t* act_i, const float* act_f, const emm_tape* tape, const emm_step_out* out, void* stream)
{
    EM_HIP(hipSetDevice(h->device));
    MarlArgs a;
    std::memset(&a, 0, sizeof a);
    a.st = h->st;
    a.kc = h->kc_dev;
    a.act_i = act_i;
    a.act_f = act_f;
    if (tape) { a.tape = *tape; a.has_tape = 1; }
    if (out) a.out = *out;
    a.n_envs = h->n_envs;
    a.env_offset = h->env_offset;
    a.key0 = (uint32_t)h->seed;
    a.key1 = (uint32_t)(h->seed >> 32);
    a.episode = h->episode;
    hipStream_t s = (hipStream_t)stream;
    if (h->lanes_per_env == 16) {
        const unsigned grid16 = (unsigned)((h->n_envs + ENVS16 - 1) / ENVS16);
        if (reset) hipLaunchKernelGGL(marl_env16_kernel<true>, dim3(grid16), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(marl_env16_kernel<false>, dim3(grid16), dim3(256), 0, s, a);
        return em::hip_check(hipGetLastError(), "EdgeMARLEnv kernel launch");
    }
    const unsigned grid = (unsigned)((h->n_envs + h->threads - 1) / h->threads);
    if (reset) hipLaunchKernelGGL(marl_env_kernel<true>, dim3(grid), dim3(h->threads), h->lds_bytes, s, a);
    else hipLaunchKernelGGL(marl_env_kernel<false>, dim3(grid), dim3(h->threads), h->lds_bytes, s, a);
    return em::hip_check(hipGetLastError(), "EdgeMARLEnv kernel launch");
}

}  // namespace

extern "C" {

int emm_create(const emm_config* cfg, int64_t n_envs, int device_id, uint64_t seed, uint64_t env_offset, emm_handle** out)
{
    if (!cfg || !out) return em::fail(EM_ERR_INVALID, "emm_create: null argument");
    if (n_envs < 1) return em::fail(EM_ERR_INVALID, "emm_create: n_envs must be positive");
    if (int rc = check_cfg(cfg)) return rc;
    EM_HIP(hipSetDevice(device_id));
    emm_handle* h = new (std::nothrow) emm_handle();
    if (!h) return em::fail(EM_ERR_INVALID, "emm_create: out of host memory");
    h->cfg = *cfg;
    h->n_envs = n_envs;
    h->device = device_id;
    h->seed = seed;
    h->env_offset = env_offset;
    h->episode = 0;
    const int n = cfg->n_nodes;
    h->lanes_per_env = pick_lanes(n, n_envs);
    // 20 bytes of LDS per (agent, lane): as many lanes per workgroup as fit 64 KB, at most 256
    const size_t per_lane = (size_t)n * 20;
    int threads = 256;
    while (threads > 64 && per_lane * (threads + 1) > 64 * 1024) threads >>= 1;
    h->threads = threads;
    h->lds_bytes = per_lane * (threads + 1);
    if (h->lds_bytes > 64 * 1024) {
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(&marl_env_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(&marl_env_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_bytes);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            const size_t need = h->lds_bytes;
            delete h;
            return em::fail(EM_ERR_INVALID, "emm_create: n_nodes %d needs %zu bytes of LDS per workgroup, which this device refuses", n, need);
        }
    }
    MarlConst kc;
    std::memset(&kc, 0, sizeof kc);
    for (int i = 0; i < EM_MAX_NODES; ++i) kc.cap[i] = i < n ? cfg->node_cpu_capacity[i] : 1.0;
    kc.rate_fixed = cfg->arrival_rate; kc.rate_lo = cfg->arrival_rate_lo; kc.rate_hi = cfg->arrival_rate_hi; kc.cycle = cfg->arrival_cycle_seconds;
    kc.wl_lo = cfg->workload_lo; kc.wl_hi = cfg->workload_hi; kc.data_lo = cfg->data_lo; kc.data_hi = cfg->data_hi;
    kc.ddl_lo = cfg->deadline_lo; kc.ddl_hi = cfg->deadline_hi; kc.bw_lo = cfg->bw_lo; kc.bw_hi = cfg->bw_hi;
    kc.lat_lo = cfg->net_latency_lo; kc.lat_hi = cfg->net_latency_hi;
    kc.alpha = cfg->alpha; kc.beta = cfg->beta; kc.gamma = cfg->gamma; kc.dt_min = cfg->min_time_step; kc.dt_max = cfg->max_time_step;
    kc.n = n; kc.limit = cfg->episode_length; kc.arrival_mode = cfg->arrival_mode; kc.action_mode = cfg->action_mode;
    hipError_t e = hipMalloc(&h->kc_dev, sizeof kc);
    if (e == hipSuccess) e = hipMemcpy(h->kc_dev, &kc, sizeof kc, hipMemcpyHostToDevice);
    // one block: load, wl, data, deadline [n][B] f64 | sim_time [B] f64 | queue [n][B] f32 | step [B] i32
    const size_t nB = (size_t)n * (size_t)n_envs, Bz = (size_t)n_envs;
    const size_t bytes = (4 * nB + Bz) * sizeof(double) + nB * sizeof(float) + Bz * sizeof(int32_t);
    if (e == hipSuccess) e = hipMalloc(&h->state_block, bytes);
    if (e == hipSuccess) e = hipMemset(h->state_block, 0, bytes);
    if (e != hipSuccess) {
        if (h->kc_dev) (void)hipFree(h->kc_dev);
        if (h->state_block) (void)hipFree(h->state_block);
        delete h;
        return em::hip_fail(e, "emm_create: device allocation");
    }
    double* d = static_cast<double*>(h->state_block);
    h->st.load = d; h->st.wl = d + nB; h->st.data = d + 2 * nB; h->st.deadline = d + 3 * nB; h->st.sim_time = d + 4 * nB;
    h->st.queue = reinterpret_cast<float*>(d + 4 * nB + Bz);
    h->st.step = reinterpret_cast<int32_t*>(h->st.queue + nB);
    *out = h;
    return EM_OK;
}

int emm_destroy(emm_handle* h)
{
    if (!h) return EM_OK;
    (void)hipSetDevice(h->device);
    (void)hipFree(h->kc_dev);
    (void)hipFree(h->state_block);
    delete h;
    return EM_OK;
}

int emm_reset(emm_handle* h, const emm_tape* tape, int has_seed, uint64_t seed, const emm_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "emm_reset: null handle");
    if (tape && (!tape->task_workload || !tape->task_data || !tape->task_deadline_offset))
        return em::fail(EM_ERR_INVALID, "emm_reset: the tape needs task_workload, task_data and task_deadline_offset");
    if (has_seed) { h->seed = seed; h->episode = 0; }
    h->episode += 1;
    return launch(h, true, nullptr, nullptr, tape, out, stream);
}

int emm_step(emm_handle* h, const int32_t* actions_i32, const float* actions_f32, const emm_tape* tape, const emm_step_out* out, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "emm_step: null handle");
    if (h->episode == 0) return em::fail(EM_ERR_STATE, "emm_step: call emm_reset first");
    if (h->cfg.action_mode == 0 && !actions_i32) return em::fail(EM_ERR_INVALID, "emm_step: discrete mode needs actions_i32");
    if (h->cfg.action_mode == 1 && !actions_f32) return em::fail(EM_ERR_INVALID, "emm_step: hybrid mode needs actions_f32");
    if (tape) {
        if (!tape->dt_raw || !tape->bw_u || !tape->net_lat || !tape->task_workload || !tape->task_data || !tape->task_deadline_offset)
            return em::fail(EM_ERR_INVALID, "emm_step: the tape is missing an array");
    }
    return launch(h, false, actions_i32, actions_f32, tape, out, stream);
}

namespace {
__global__ void marl_get_state_kernel(MarlState st, int n, int64_t B, double* load, double* queue, double* sim_time, int32_t* step)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    for (int j = 0; j < n; ++j) {
        if (load) load[b * n + j] = st.load[(int64_t)j * B + b];
        if (queue) queue[b * n + j] = (double)st.queue[(int64_t)j * B + b];
    }
    if (sim_time) sim_time[b] = st.sim_time[b];
    if (step) step[b] = st.step[b];
}
}  // namespace

int emm_set_lanes_per_env(emm_handle* h, int32_t lanes_per_env)
{
    if (!h) return em::fail(EM_ERR_INVALID, "emm_set_lanes_per_env: null handle");
    if (lanes_per_env == 0) { h->lanes_per_env = pick_lanes(h->cfg.n_nodes, h->n_envs); return EM_OK; }
    if (lanes_per_env != 1 && lanes_per_env != 16) return em::fail(EM_ERR_INVALID, "emm_set_lanes_per_env: 0 (automatic), 1 or 16");
    if (lanes_per_env == 16 && h->cfg.n_nodes > G16) return em::fail(EM_ERR_INVALID, "emm_set_lanes_per_env: 16 lanes need n_nodes <= 16");
    h->lanes_per_env = lanes_per_env;
    return EM_OK;
}

int emm_get_state(emm_handle* h, double* load, double* queue, double* sim_time, int32_t* step, void* stream)
{
    if (!h) return em::fail(EM_ERR_INVALID, "emm_get_state: null handle");
    EM_HIP(hipSetDevice(h->device));
    hipLaunchKernelGGL(marl_get_state_kernel, dim3((unsigned)((h->n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, h->st,
                       h->cfg.n_nodes, h->n_envs, load, queue, sim_time, step);
    return em::hip_check(hipGetLastError(), "emm_get_state launch");
}

}  // extern "C"
