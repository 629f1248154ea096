// scaled_rollout.hip - trajectory indexing for the scaled topology (RolloutBuffer.add / MAPPOTrainer._store_transition,
// mappo_algorithm.py:104-111, mappo_trainer.py:160-175, fed by the COMPACT pending lists of ems_step).
//
// Per step: (1) entries (acting (env, decider) transitions) are numbered in (environment, arrival slot) order by an
// exclusive scan of the per-environment pending counts; (2) the step's entries are grouped by the network that
// serves them (decider id for independent learners, one group for a shared network) with a counting sort, which gives
// the group offsets and the position -> entry permutation the grouped MLP kernels (gmlp.hip) consume; (3) the
// observation rows are copied into the entry storage.  (1)+(2) run in ONE workgroup (the work is a few items per
// environment), (3) is a wide copy kernel.  The order inside a group comes from LDS atomics and may vary from run to
// run: the rollout forward treats rows independently, so results do not depend on it (the update sorts stably).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "edgemarl.h"
#include "em_common.h"

namespace {

constexpr int PCAP = EMS_PENDING_CAP;
constexpr int IDX_THREADS = 1024;

__global__ __launch_bounds__(IDX_THREADS) void scaled_step_index_kernel(const ems_rollout ro, int t, int n_groups,
                                                                        const int32_t* __restrict__ n_pending,
                                                                        const int32_t* __restrict__ pend_decider)
{
    extern __shared__ int32_t sm[];
    int32_t* hist = sm;                       // [n_groups] counts, then running cursors
    int32_t* part = sm + n_groups;            // [IDX_THREADS] partial sums
    const int tid = threadIdx.x;
    const int64_t B = ro.n_envs;
    const int64_t base = ro.counters[t];
    for (int i = tid; i < n_groups; i += IDX_THREADS) hist[i] = 0;
    __syncthreads();

    // ---- (1) entries per environment: chunked exclusive scan; group histogram on the way ---------------------
    const int64_t chunk = (B + IDX_THREADS - 1) / IDX_THREADS;
    const int64_t b0 = (int64_t)tid * chunk, b1 = b0 + chunk < B ? b0 + chunk : B;
    int32_t local = 0;
    for (int64_t b = b0; b < b1; ++b) {
        int k = n_pending[b];
        k = k < 0 ? 0 : (k > PCAP ? PCAP : k);
        local += k;
        if (n_groups > 1)
            for (int j = 0; j < k; ++j) atomicAdd(&hist[pend_decider[b * PCAP + j]], 1);
    }
    part[tid] = local;
    __syncthreads();
    for (int d = 1; d < IDX_THREADS; d <<= 1) {         // inclusive Hillis-Steele scan of the partial sums
        const int32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    const int32_t total = part[IDX_THREADS - 1];
    const bool overflow = base + total > ro.capacity_entries;
    int32_t run = part[tid] - local;
    uint8_t* k_row = ro.k_tb + (int64_t)t * B;
    int32_t* f_row = ro.first_tb + (int64_t)t * B;
    for (int64_t b = b0; b < b1; ++b) {
        int k = n_pending[b];
        k = k < 0 ? 0 : (k > PCAP ? PCAP : k);
        k_row[b] = overflow ? 0 : (uint8_t)k;
        f_row[b] = (int32_t)(base + run);
        run += k;
    }
    if (tid == 0) {
        if (overflow) atomicOr(ro.err_flag, 8);
        ro.counters[t + 1] = base + (overflow ? 0 : total);
        if (n_groups == 1) { ro.group_off[0] = 0; ro.group_off[1] = overflow ? 0 : total; }
    }
    if (overflow) {
        for (int i = tid; i <= n_groups + 2; i += IDX_THREADS) ro.group_off[i] = 0;
        return;
    }
    if (n_groups == 1) {                                  // one shared network: identity order
        run = part[tid] - local;
        for (int64_t b = b0; b < b1; ++b) {
            const int k = k_row[b];
            for (int j = 0; j < k; ++j) ro.perm[run + j] = (int32_t)(base + run + j);
            run += k;
        }
        return;
    }
    __syncthreads();

    // ---- (2) exclusive scan of the histogram -> group offsets and cursors --------------------------------------
    const int gchunk = (n_groups + IDX_THREADS - 1) / IDX_THREADS;
    const int g0 = tid * gchunk, g1 = g0 + gchunk < n_groups ? g0 + gchunk : n_groups;
    int32_t gl = 0;
    for (int g = g0; g < g1; ++g) gl += hist[g];
    __syncthreads();
    part[tid] = gl;
    __syncthreads();
    for (int d = 1; d < IDX_THREADS; d <<= 1) {
        const int32_t v = tid >= d ? part[tid - d] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int32_t grun = part[tid] - gl;
    for (int g = g0; g < g1; ++g) {
        const int32_t c = hist[g];
        ro.group_off[g] = grun;
        hist[g] = grun;                                   // cursor
        grun += c;
    }
    if (tid == 0) {
        ro.group_off[n_groups] = total;
        ro.group_off[n_groups + 1] = 0;       // {0, total}: the whole step as ONE group, for a shared network that runs
        ro.group_off[n_groups + 2] = total;   // over the same permutation (any order of the step's entries serves it)
    }
    __syncthreads();

    // ---- (3) positions: entry (first[b] + slot) goes to the next free position of its group ----------------------
    for (int64_t b = b0; b < b1; ++b) {
        const int k = k_row[b];
        const int32_t f = f_row[b];
        for (int j = 0; j < k; ++j) {
            const int pos = atomicAdd(&hist[pend_decider[b * PCAP + j]], 1);
            ro.perm[pos] = f + j;
        }
    }
}

// entry rows of step t: observation, environment, decider, destination slot of the action (env * 64 + slot)
__global__ __launch_bounds__(256) void scaled_step_gather_kernel(const ems_rollout ro, int t, const int32_t* __restrict__ pend_decider,
                                                                 const float* __restrict__ pend_obs)
{
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= ro.n_envs) return;
    const int k = ro.k_tb[(int64_t)t * ro.n_envs + b];
    if (k == 0) return;
    const int64_t f = ro.first_tb[(int64_t)t * ro.n_envs + b];
    const int w = ro.n_servers + 4;
    const float* src = pend_obs + b * PCAP * (int64_t)w;
    float* dst = ro.ent_obs + f * w;
    for (int i = lane; i < k * w; i += 64) dst[i] = src[i];      // k consecutive rows on both sides
    if (lane < k) {
        ro.ent_env[f + lane] = (int32_t)b;
        ro.ent_agent[f + lane] = pend_decider[b * PCAP + lane];
        ro.ent_dst[f + lane] = (int32_t)(b * PCAP + lane);
        if (ro.ent_state) ro.ent_state[f + lane] = (int32_t)((int64_t)t * ro.n_envs + b);
    }
}

}  // namespace

extern "C" {

int ems_rollout_index(const ems_rollout* ro, int32_t t, int32_t n_groups, const int32_t* n_pending, const int32_t* pend_decider,
                      const float* pend_obs, void* stream)
{
    if (!ro || !n_pending || !pend_decider || !pend_obs) return em::fail(EM_ERR_INVALID, "ems_rollout_index: null argument");
    if (t < 0 || t >= ro->horizon) return em::fail(EM_ERR_INVALID, "ems_rollout_index: step %d outside [0,%d)", t, ro->horizon);
    if (n_groups != 1 && n_groups != ro->n_deciders) return em::fail(EM_ERR_INVALID, "ems_rollout_index: n_groups must be 1 or n_deciders");
    if (ro->n_deciders < 1 || ro->n_deciders > EMS_MAX_DECIDERS) return em::fail(EM_ERR_INVALID, "ems_rollout_index: bad n_deciders");
    if (!ro->ent_obs || !ro->ent_env || !ro->ent_agent || !ro->ent_dst || !ro->k_tb || !ro->first_tb || !ro->counters ||
        !ro->group_off || !ro->perm || !ro->err_flag)
        return em::fail(EM_ERR_INVALID, "ems_rollout_index: storage pointers missing");
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(int32_t) * ((size_t)n_groups + IDX_THREADS);
    hipLaunchKernelGGL(scaled_step_index_kernel, dim3(1), dim3(IDX_THREADS), lds, s, *ro, t, n_groups, n_pending, pend_decider);
    hipLaunchKernelGGL(scaled_step_gather_kernel, dim3((unsigned)((ro->n_envs + 3) / 4)), dim3(256), 0, s, *ro, t, pend_decider,
                       pend_obs);
    return em::hip_check(hipGetLastError(), "scaled rollout index launch");
}

}  // extern "C"
