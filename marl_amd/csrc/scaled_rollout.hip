// scaled_rollout.hip - trajectory indexing for the scaled topology (RolloutBuffer.add / MAPPOTrainer._store_transition,
// mappo_algorithm.py:104-111, mappo_trainer.py:160-175, fed by the COMPACT pending lists of ems_step).
//
// Per step: (1) entries (acting (env, decider) transitions) are numbered in (environment, arrival slot) order by an
// exclusive scan of the per-environment pending counts; (2) the step's entries are grouped by the network that
// serves them (decider id for independent learners, one group for a shared network) with a counting sort, which gives
// the group offsets and the position -> entry permutation the grouped MLP kernels (gmlp.hip) consume; (3) the
// observation rows are copied into the entry storage (or left to the step's em_gmlp_policy_gather launch, which reads them
// anyway: pend_obs == NULL).  Two short launches: a chunk-parallel count (scan inside
// 256-environment chunks + histogram) and a wavefront-per-environment placement + copy in which every wavefront redoes
// the scan of the chunk totals and of the histogram for itself (scaled_index_place2_kernel); above 256 groups, or with
// EMS_INDEX_THREE_LAUNCHES=1, a single-workgroup scan launch stands between the two.  The order inside a group comes from
// integer atomics and may vary from run to run: the rollout forward treats rows independently, so results do not depend
// on it (the update sorts stably).
// (A single-workgroup version of (1)+(2) cost 29 us per step at 4,096 environments and 117 us at 16,384.)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "edgemarl.h"
#include "em_common.h"

namespace {

constexpr int PCAP = EMS_PENDING_CAP;
constexpr int CHUNK = 256;            // environments per workgroup of the count kernel
constexpr int CUR_STRIDE = EMS_INDEX_CURSOR_STRIDE;   // one 128-byte line per group cursor: the ~140 atomics a cursor takes per
                                                      // step then spread over the L2 channels (8 shared lines cost 45 us at 16,384 envs)

// (1) per 256-environment chunk: entries per environment, exclusive scan inside the chunk, chunk total, and the chunk's
// contribution to the per-group histogram (LDS histogram, flushed with integer atomics: exact in any order).
__global__ __launch_bounds__(CHUNK) void scaled_index_count_kernel(const ems_rollout ro, int t, int n_groups,
                                                                   const int32_t* __restrict__ n_pending,
                                                                   const int32_t* __restrict__ pend_decider,
                                                                   int32_t* __restrict__ chunk_tot, int32_t* __restrict__ hist)
{
    extern __shared__ int32_t lh[];               // [n_groups] when grouping
    __shared__ int32_t wsum[CHUNK / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t B = ro.n_envs, b = (int64_t)blockIdx.x * CHUNK + tid;
    if (n_groups > 1) {
        for (int i = tid; i < n_groups; i += CHUNK) lh[i] = 0;
        __syncthreads();
    }
    // the count and the first four deciders of the list are requested together (2.2 entries per environment on average):
    // one memory round trip instead of two for nearly every environment
    const int64_t bc = b < B ? b : B - 1;
    int k = n_pending[bc];
    const int4 d4 = *reinterpret_cast<const int4*>(pend_decider + bc * PCAP);
    k = (b < B) ? (k < 0 ? 0 : (k > PCAP ? PCAP : k)) : 0;
    if (n_groups > 1) {
        if (k > 0) atomicAdd(&lh[d4.x], 1);
        if (k > 1) atomicAdd(&lh[d4.y], 1);
        if (k > 2) atomicAdd(&lh[d4.z], 1);
        if (k > 3) atomicAdd(&lh[d4.w], 1);
        for (int j = 4; j < k; ++j) atomicAdd(&lh[pend_decider[b * PCAP + j]], 1);
    }
    int inc = k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    int woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < CHUNK / 64; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
    if (b < B) {
        ro.k_tb[(int64_t)t * B + b] = (uint8_t)k;
        ro.first_tb[(int64_t)t * B + b] = woff + inc - k;        // offset inside the chunk; the place kernel adds the bases
    }
    if (tid == 0) chunk_tot[blockIdx.x] = tot;
    if (n_groups > 1)
        for (int i = tid; i < n_groups; i += CHUNK) { const int c = lh[i]; if (c) atomicAdd(&hist[i], c); }
}

// (2) one workgroup: chunk bases, the step's total, capacity check, group offsets and cursors; re-zeroes the histogram
__global__ __launch_bounds__(256) void scaled_index_scan_kernel(const ems_rollout ro, int t, int n_groups, int n_chunks,
                                                                int32_t* __restrict__ chunk_tot, int32_t* __restrict__ hist,
                                                                int32_t* __restrict__ cursor, int32_t* __restrict__ ok_flag)
{
    __shared__ int32_t part[256];
    __shared__ int32_t s_total;
    const int tid = threadIdx.x;
    auto block_scan = [&](int32_t v) -> int32_t {          // inclusive scan of one value per thread
        part[tid] = v;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const int32_t o = tid >= d ? part[tid - d] : 0;
            __syncthreads();
            part[tid] += o;
            __syncthreads();
        }
        const int32_t r = part[tid];
        __syncthreads();
        return r;
    };
    // chunk totals -> exclusive bases (in place)
    const int cper = (n_chunks + 255) / 256;
    const int c0 = tid * cper, c1 = c0 + cper < n_chunks ? c0 + cper : n_chunks;
    int32_t loc = 0;
    for (int i = c0; i < c1; ++i) loc += chunk_tot[i];
    int32_t run = block_scan(loc) - loc;
    for (int i = c0; i < c1; ++i) { const int32_t v = chunk_tot[i]; chunk_tot[i] = run; run += v; }
    if (tid == 255) s_total = run;
    __syncthreads();
    const int32_t total = s_total;
    const int64_t base = ro.counters[t];
    const bool overflow = base + total > ro.capacity_entries;
    if (tid == 0) {
        if (overflow) atomicOr(ro.err_flag, 8);
        ro.counters[t + 1] = base + (overflow ? 0 : total);
        *ok_flag = overflow ? 0 : 1;
    }
    const int32_t eff = overflow ? 0 : total;
    if (n_groups == 1) {
        if (tid == 0) { ro.group_off[0] = 0; ro.group_off[1] = eff; }
        return;
    }
    const int gper = (n_groups + 255) / 256;
    const int g0 = tid * gper, g1 = g0 + gper < n_groups ? g0 + gper : n_groups;
    int32_t gl = 0;
    for (int g = g0; g < g1; ++g) gl += hist[g];
    int32_t grun = block_scan(gl) - gl;
    for (int g = g0; g < g1; ++g) {
        const int32_t c = hist[g];
        ro.group_off[g] = overflow ? 0 : grun;
        cursor[g * CUR_STRIDE] = grun;
        hist[g] = 0;                                           // ready for the next step
        grun += c;
    }
    if (tid == 0) {
        ro.group_off[n_groups] = eff;
        ro.group_off[n_groups + 1] = 0;       // {0, total}: the step as ONE group, for a shared network that runs over
        ro.group_off[n_groups + 2] = eff;     // the same permutation (any order of the step's entries serves it)
    }
}

// (3) one wavefront per environment: final entry numbers, positions inside the groups (integer atomics on the
// cursors: the order inside a group may vary, which cannot change a forward result), entry rows
__global__ __launch_bounds__(256) void scaled_index_place_kernel(const ems_rollout ro, int t, int n_groups,
                                                                 const int32_t* __restrict__ pend_decider,
                                                                 const float* __restrict__ pend_obs,
                                                                 const int32_t* __restrict__ chunk_base, int32_t* __restrict__ cursor,
                                                                 const int32_t* __restrict__ ok_flag)
{
    const int lane = threadIdx.x & 63;
    const int64_t B = ro.n_envs, b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t base = ro.counters[t];
    const int64_t i = (int64_t)t * B + b;
    const int32_t rel = chunk_base[b / CHUNK] + ro.first_tb[i];          // position of the first entry inside the step
    const int64_t f = base + rel;
    int k = ro.k_tb[i];
    if (*ok_flag == 0) {                                                  // capacity overflow: the step records nothing
        if (lane == 0) { ro.k_tb[i] = 0; ro.first_tb[i] = (int32_t)base; }
        return;
    }
    if (lane == 0) ro.first_tb[i] = (int32_t)f;
    if (k == 0) return;
    const int w = ro.n_servers + 4;
    const float* src = pend_obs + b * PCAP * (int64_t)w;
    float* dst = ro.ent_obs + f * w;
    if (pend_obs)
        for (int x = lane; x < k * w; x += 64) dst[x] = src[x];           // k consecutive rows on both sides
    if (lane < k) {
        const int dec = pend_decider[b * PCAP + lane];
        const int pos = n_groups > 1 ? atomicAdd(&cursor[dec * CUR_STRIDE], 1) : rel + lane;
        ro.perm[pos] = (int32_t)(f + lane);
        ro.ent_env[f + lane] = (int32_t)b;
        ro.ent_agent[f + lane] = dec;
        ro.ent_dst[f + lane] = (int32_t)(b * PCAP + lane);
        if (ro.ent_state) ro.ent_state[f + lane] = (int32_t)i;
    }
}


// (2)+(3) in one launch, for n_groups == 1 or up to PLACE2_MAX_GROUPS groups: every wavefront (two environments, 32 lanes
// each: a list holds 2.2 entries on average) sums the chunk totals in front of its chunk and all of them (64 values at
// 16,384 environments), and - when one of its environments has entries - scans the step's group histogram itself (four
// counts per lane at 256 groups) instead of reading what a single-workgroup launch in between would have written: one
// launch and one kernel boundary fewer per step.  The histogram and the cursors are double-buffered by step parity: this
// launch reads the buffers of step t and the wavefront of environment 0 clears those of step t + 1, which no launch in
// flight uses; that wavefront also writes the step's counters and group offsets.  Cursors start at zero and are offsets
// INSIDE the group.
// Every load of the prologue is issued up front, unconditionally and on addresses that are valid for any list length (a
// load under `if (k > 0)` waits for k first: four dependent memory round trips per wavefront became two): chunk totals,
// counters, the environments' counts and offsets, the histogram, the first 32 deciders of each list and the first
// PRE x 32 floats of its observation rows; longer lists take the loops.  (Copying the rows in 16-byte pieces was SLOWER:
// 21-24 us instead of 11-14 us per step, whatever the number of lanes per environment.)  What is left of the 11 us is the
// copy itself: 10 MB read + 10 MB written per step.
constexpr int PLACE2_MAX_GROUPS = 256;
constexpr int PLACE2_SUB = 32;                   // lanes per environment (16: 11.5 us, 32: 11.2 us, 64: 13.7 us per step at 16,384 environments)
constexpr int PLACE2_EPW = 64 / PLACE2_SUB;      // environments per wavefront (divides CHUNK: one chunk per wavefront)
__global__ __launch_bounds__(256) void scaled_index_place2_kernel(const ems_rollout ro, int t, int n_groups, int n_chunks,
                                                                  const int32_t* __restrict__ pend_decider,
                                                                  const float* __restrict__ pend_obs,
                                                                  const int32_t* __restrict__ chunk_tot,
                                                                  const int32_t* __restrict__ hist, int32_t* __restrict__ cursor,
                                                                  int32_t* __restrict__ hist_next, int32_t* __restrict__ cursor_next)
{
    __shared__ int32_t goff[4][PLACE2_MAX_GROUPS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane / PLACE2_SUB, sl = lane % PLACE2_SUB;
    const int64_t B = ro.n_envs, b0 = ((int64_t)blockIdx.x * 4 + wave) * PLACE2_EPW, b = b0 + sub;
    if (b0 >= B) return;                                  // whole wavefronts leave; nothing below synchronises across wavefronts
    const bool live = b < B;
    const int64_t bc = live ? b : B - 1;
    const int chunk = (int)(b0 / CHUNK);
    constexpr int PRE = 3;
    const int64_t i = (int64_t)t * B + bc;
    const int w = ro.n_servers + 4;
    const float* src = pend_obs + bc * PCAP * (int64_t)w;
    const int32_t ct0 = lane < n_chunks ? chunk_tot[lane] : 0;
    const int64_t base = ro.counters[t];
    const int k = live ? (int)ro.k_tb[i] : 0;
    const int32_t first = ro.first_tb[i];
    int32_t h[4] = {0, 0, 0, 0};
    if (n_groups > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int g = 4 * lane + j; h[j] = hist[g < n_groups ? g : 0]; }
    }
    const int dec0 = pend_decider[bc * PCAP + sl];
    const bool copy = pend_obs != nullptr;                // NULL: the caller's policy launch copies the rows (em_gmlp_policy_gather)
    float ob[PRE] = {};
    if (copy) {
#pragma unroll
        for (int j = 0; j < PRE; ++j) ob[j] = src[sl + PLACE2_SUB * j];   // PRE * 32 <= PCAP * w: inside the environment's rows
    }
    int32_t pre = chunk > lane ? ct0 : 0, all = ct0;
    for (int c = lane + 64; c < n_chunks; c += 64) {
        const int32_t v = chunk_tot[c];
        all += v;
        if (c < chunk) pre += v;
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { pre += __shfl_xor(pre, d); all += __shfl_xor(all, d); }
    const int32_t rel = pre + first;                                        // position of the first entry inside the step
    const bool overflow = base + all > ro.capacity_entries;
    const int32_t eff = overflow ? 0 : all;
    const bool book = b0 == 0;                            // this wavefront keeps the step's books
    if (book && lane == 0) {
        if (overflow) atomicOr(ro.err_flag, 8);
        ro.counters[t + 1] = base + eff;
        if (n_groups == 1) { ro.group_off[0] = 0; ro.group_off[1] = eff; }
        else {
            ro.group_off[n_groups] = eff;
            ro.group_off[n_groups + 1] = 0;       // {0, total}: the step as ONE group, for a shared network that runs over
            ro.group_off[n_groups + 2] = eff;     // the same permutation (any order of the step's entries serves it)
        }
    }
    if (n_groups > 1 && (book || (__ballot(k > 0) != 0ull && !overflow))) {
        // exclusive scan of the histogram: lane l holds groups 4l .. 4l + 3
        int32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (4 * lane + j >= n_groups) h[j] = 0; sum += h[j]; }
        int32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int32_t o = __shfl_up(inc, d);
            if (lane >= d) inc += o;
        }
        int32_t run = inc - sum;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 4 * lane + j;
            goff[wave][g] = run;
            if (book && g < n_groups) {
                ro.group_off[g] = overflow ? 0 : run;
                hist_next[g] = 0;
                cursor_next[g * CUR_STRIDE] = 0;
            }
            run += h[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (!live) return;
    if (overflow) {                                                       // capacity overflow: the step records nothing
        if (sl == 0) { ro.k_tb[i] = 0; ro.first_tb[i] = (int32_t)base; }
        return;
    }
    const int64_t f = base + rel;
    if (sl == 0) ro.first_tb[i] = (int32_t)f;
    if (k == 0) return;
    float* dst = ro.ent_obs + f * w;
    const int np = k * w;                                                 // k consecutive rows on both sides
    if (copy) {
#pragma unroll
        for (int j = 0; j < PRE; ++j)
            if (sl + PLACE2_SUB * j < np) dst[sl + PLACE2_SUB * j] = ob[j];
        for (int x = sl + PLACE2_SUB * PRE; x < np; x += PLACE2_SUB) dst[x] = src[x];
    }
    for (int slot = sl; slot < k; slot += PLACE2_SUB) {
        const int dec = slot == sl ? dec0 : pend_decider[b * PCAP + slot];
        const int pos = n_groups > 1 ? goff[wave][dec] + atomicAdd(&cursor[dec * CUR_STRIDE], 1) : rel + slot;
        ro.perm[pos] = (int32_t)(f + slot);
        ro.ent_env[f + slot] = (int32_t)b;
        ro.ent_agent[f + slot] = dec;
        ro.ent_dst[f + slot] = (int32_t)(b * PCAP + slot);
        if (ro.ent_state) ro.ent_state[f + slot] = (int32_t)i;
    }
}


// ------------------------------------------------------------------------------------------------
// Grouping of a whole batch of entries by serving network for the update (the reference has one PPO learner per agent:
// MRRL.PY:611-638 trains them independently, so a learner's rows must be contiguous and in a reproducible order).  A STABLE
// counting sort in three launches - rows of a group keep their storage order, so every learner sums its rows in the same
// order from run to run - replacing torch.sort's eight radix passes over 64-bit keys (4.5 ms per update at 36.7 M rows):
//   hist   one wavefront per chunk of GI_CHUNK rows: the chunk's histogram (LDS integer atomics: exact in any order)
//   scan   exclusive prefix of every group's counts over the chunks (segments of chunks first, then the segments), and
//          the group offsets
//   place  one wavefront per chunk again, 64 rows at a time in storage order: lanes with the same key find each other
//          with one ballot per key bit, rank themselves by lane number and take consecutive positions from the group's
//          running offset in LDS
// ------------------------------------------------------------------------------------------------
constexpr int GI_CHUNK = 2048;        // rows per wavefront
constexpr int GI_SEG = 64;            // chunks are scanned in this many segments

__global__ __launch_bounds__(64) void group_hist_kernel(const int32_t* __restrict__ key, int64_t n, int G, int32_t* __restrict__ hist)
{
    extern __shared__ int32_t cnt[];                // [G]
    const int lane = threadIdx.x;
    for (int i = lane; i < G; i += 64) cnt[i] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * GI_CHUNK;
    for (int i = lane; i < GI_CHUNK; i += 64) {
        const int64_t r = r0 + i;
        if (r < n) {
            int k = key[r];
            k = k < 0 ? 0 : (k >= G ? G - 1 : k);
            atomicAdd(&cnt[k], 1);
        }
    }
    __syncthreads();
    for (int i = lane; i < G; i += 64) hist[(int64_t)blockIdx.x * G + i] = cnt[i];
}

// workgroup s: group g's counts of segment s's chunks -> exclusive prefix inside the segment (in place), segment total
__global__ __launch_bounds__(256) void group_scan_seg_kernel(int32_t* __restrict__ hist, int n_chunks, int G, int per_seg,
                                                             int32_t* __restrict__ seg_tot)
{
    const int c0 = blockIdx.x * per_seg, c1 = c0 + per_seg < n_chunks ? c0 + per_seg : n_chunks;
    for (int g = threadIdx.x; g < G; g += 256) {
        int32_t run = 0;
        for (int c = c0; c < c1; ++c) {
            const int32_t v = hist[(int64_t)c * G + g];
            hist[(int64_t)c * G + g] = run;
            run += v;
        }
        seg_tot[blockIdx.x * G + g] = run;
    }
}

// one workgroup: per group the exclusive prefix over the segments (in place) and the group offsets
__global__ __launch_bounds__(1024) void group_scan_top_kernel(int32_t* __restrict__ seg_tot, int n_seg, int G, int64_t* __restrict__ group_off)
{
    __shared__ long long tot[1024];
    const int g = threadIdx.x;
    long long run = 0;
    if (g < G) {
        for (int s = 0; s < n_seg; ++s) {
            const int32_t v = seg_tot[s * G + g];
            seg_tot[s * G + g] = (int32_t)run;
            run += v;
        }
    }
    tot[g] = g < G ? run : 0;
    __syncthreads();
    if (g == 0) {
        long long acc = 0;
        for (int i = 0; i < G; ++i) { group_off[i] = acc; acc += tot[i]; }
        group_off[G] = acc;
    }
}

__global__ __launch_bounds__(64) void group_place_kernel(const int32_t* __restrict__ key, int64_t n, int G, int n_bits, int per_seg,
                                                         const int32_t* __restrict__ hist, const int32_t* __restrict__ seg_tot,
                                                         const int64_t* __restrict__ group_off, int32_t* __restrict__ perm)
{
    extern __shared__ int32_t off[];                // [G] next position of each group, for this chunk
    const int lane = threadIdx.x;
    const int seg = blockIdx.x / per_seg;
    for (int i = lane; i < G; i += 64)
        off[i] = (int32_t)group_off[i] + seg_tot[seg * G + i] + hist[(int64_t)blockIdx.x * G + i];
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * GI_CHUNK;
    for (int i = 0; i < GI_CHUNK; i += 64) {
        const int64_t r = r0 + i + lane;
        const bool valid = r < n;
        if (__ballot(valid) == 0ull) break;
        int k = valid ? key[r] : 0;
        k = k < 0 ? 0 : (k >= G ? G - 1 : k);
        unsigned long long m = __ballot(valid);                        // lanes of this pass that hold the same key
        for (int b = 0; b < n_bits; ++b) {
            const bool bit = (k >> b) & 1;
            const unsigned long long bal = __ballot(valid && bit);
            m &= bit ? bal : ~bal;
        }
        const int base = valid ? off[k] : 0;                            // read by every lane of the key before its leader adds
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const int rank = __popcll(m & ((1ull << lane) - 1ull));
            if (rank == 0) off[k] = base + __popcll(m);
            perm[(int64_t)base + rank] = (int32_t)r;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

extern "C" {

int ems_rollout_index(const ems_rollout* ro, int32_t t, int32_t n_groups, const int32_t* n_pending, const int32_t* pend_decider,
                      const float* pend_obs, void* stream)
{
    if (!ro || !n_pending || !pend_decider) return em::fail(EM_ERR_INVALID, "ems_rollout_index: null argument");
    if (t < 0 || t >= ro->horizon) return em::fail(EM_ERR_INVALID, "ems_rollout_index: step %d outside [0,%d)", t, ro->horizon);
    if (n_groups != 1 && n_groups != ro->n_deciders) return em::fail(EM_ERR_INVALID, "ems_rollout_index: n_groups must be 1 or n_deciders");
    if (ro->n_deciders < 1 || ro->n_deciders > EMS_MAX_DECIDERS) return em::fail(EM_ERR_INVALID, "ems_rollout_index: bad n_deciders");
    if (!ro->ent_obs || !ro->ent_env || !ro->ent_agent || !ro->ent_dst || !ro->k_tb || !ro->first_tb || !ro->counters ||
        !ro->group_off || !ro->perm || !ro->err_flag)
        return em::fail(EM_ERR_INVALID, "ems_rollout_index: storage pointers missing");
    if (!ro->index_scratch) return em::fail(EM_ERR_INVALID, "ems_rollout_index: index_scratch is required");
    if (reinterpret_cast<uintptr_t>(pend_decider) & 15) return em::fail(EM_ERR_INVALID, "ems_rollout_index: pend_decider must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int n_chunks = (int)((ro->n_envs + CHUNK - 1) / CHUNK);
    const int D = ro->n_deciders;
    // scratch (EMS_INDEX_SCRATCH_INTS): [n_chunks] chunk totals/bases | [2][n_deciders] histograms | ok flag | pad to a
    // 128-byte line | [2][n_deciders] cursors, CUR_STRIDE ints apart.  The second histogram / cursor set serves the two-launch
    // path (buffers alternate by step parity); the three-launch path uses the first set only.
    int32_t* chunk_tot = ro->index_scratch;
    int32_t* hist = chunk_tot + n_chunks;
    int32_t* ok_flag = hist + 2 * D;
    int32_t* cursor = ro->index_scratch + ((n_chunks + 2 * D + 1 + CUR_STRIDE - 1) / CUR_STRIDE) * CUR_STRIDE;
    const size_t lds = n_groups > 1 ? sizeof(int32_t) * (size_t)n_groups : 0;
    const char* three = getenv("EMS_INDEX_THREE_LAUNCHES");
    if ((n_groups == 1 || n_groups <= PLACE2_MAX_GROUPS) && !(three && three[0] == '1')) {
        const int par = t & 1;
        if (t == 0 && n_groups > 1) {      // a new episode: whatever the last one left in either set
            EM_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * 2 * (size_t)D, s));
            EM_HIP(hipMemsetAsync(cursor, 0, sizeof(int32_t) * 2 * (size_t)D * CUR_STRIDE, s));
        }
        int32_t* hist_t = hist + par * D;
        int32_t* cur_t = cursor + (size_t)par * D * CUR_STRIDE;
        hipLaunchKernelGGL(scaled_index_count_kernel, dim3((unsigned)n_chunks), dim3(CHUNK), lds, s, *ro, t, n_groups, n_pending,
                           pend_decider, chunk_tot, hist_t);
        const dim3 grid((unsigned)((ro->n_envs + 4 * PLACE2_EPW - 1) / (4 * PLACE2_EPW)));
        hipLaunchKernelGGL(scaled_index_place2_kernel, grid, dim3(256), 0, s, *ro, t, n_groups, n_chunks, pend_decider, pend_obs,
                           chunk_tot, hist_t, cur_t, hist + (par ^ 1) * D, cursor + (size_t)(par ^ 1) * D * CUR_STRIDE);
        return em::hip_check(hipGetLastError(), "scaled rollout index launch");
    }
    if (t == 0 && n_groups > 1) EM_HIP(hipMemsetAsync(hist, 0, sizeof(int32_t) * (size_t)D, s));   // (the other path may have used it)
    hipLaunchKernelGGL(scaled_index_count_kernel, dim3((unsigned)n_chunks), dim3(CHUNK), lds, s, *ro, t, n_groups, n_pending,
                       pend_decider, chunk_tot, hist);
    hipLaunchKernelGGL(scaled_index_scan_kernel, dim3(1), dim3(256), 0, s, *ro, t, n_groups, n_chunks, chunk_tot, hist, cursor, ok_flag);
    hipLaunchKernelGGL(scaled_index_place_kernel, dim3((unsigned)((ro->n_envs + 3) / 4)), dim3(256), 0, s, *ro, t, n_groups,
                       pend_decider, pend_obs, chunk_tot, cursor, ok_flag);
    return em::hip_check(hipGetLastError(), "scaled rollout index launch");
}


int64_t em_group_index_scratch_ints(int64_t n_rows, int32_t n_groups)
{
    if (n_rows < 0 || n_groups < 1) return 0;
    const int64_t chunks = (n_rows + GI_CHUNK - 1) / GI_CHUNK;
    return (chunks + GI_SEG) * (int64_t)n_groups;
}

/* Stable grouping of n_rows rows by key (the network that serves them): see the kernels above and edgemarl.h. */
int em_group_index(const int32_t* key, int64_t n_rows, int32_t n_groups, int64_t* group_off, int32_t* perm, int32_t* scratch,
                   void* stream)
{
    if (!group_off || n_rows < 0 || n_groups < 1 || n_groups > 1024) return em::fail(EM_ERR_INVALID, "em_group_index: bad argument (1..1024 groups)");
    if (n_rows > 0x7fffffffll) return em::fail(EM_ERR_INVALID, "em_group_index: more than 2^31 - 1 rows");
    hipStream_t s = (hipStream_t)stream;
    if (n_rows == 0) {
        const hipError_t e = hipMemsetAsync(group_off, 0, sizeof(int64_t) * (size_t)(n_groups + 1), s);
        return e == hipSuccess ? EM_OK : em::hip_fail(e, "em_group_index memset");
    }
    if (!key || !perm || !scratch) return em::fail(EM_ERR_INVALID, "em_group_index: null argument");
    const int n_chunks = (int)((n_rows + GI_CHUNK - 1) / GI_CHUNK);
    const int per_seg = (n_chunks + GI_SEG - 1) / GI_SEG, n_seg = (n_chunks + per_seg - 1) / per_seg;
    int n_bits = 0;
    while ((1 << n_bits) < n_groups) ++n_bits;
    int32_t* hist = scratch;
    int32_t* seg_tot = scratch + (int64_t)n_chunks * n_groups;
    const size_t lds = sizeof(int32_t) * (size_t)n_groups;
    hipLaunchKernelGGL(group_hist_kernel, dim3((unsigned)n_chunks), dim3(64), lds, s, key, n_rows, (int)n_groups, hist);
    hipLaunchKernelGGL(group_scan_seg_kernel, dim3((unsigned)n_seg), dim3(256), 0, s, hist, n_chunks, (int)n_groups, per_seg, seg_tot);
    hipLaunchKernelGGL(group_scan_top_kernel, dim3(1), dim3(1024), 0, s, seg_tot, n_seg, (int)n_groups, group_off);
    hipLaunchKernelGGL(group_place_kernel, dim3((unsigned)n_chunks), dim3(64), lds, s, key, n_rows, (int)n_groups, n_bits, per_seg,
                       hist, seg_tot, group_off, perm);
    return em::hip_check(hipGetLastError(), "em_group_index launch");
}

}  // extern "C"
