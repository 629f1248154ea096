/*
 * edgemarl.h - C ABI of libedgemarl.so: the MI355X (gfx950) implementation of the batched
 * EdgeSimGym step and the MAPPO rollout / GAE / PPO-update kernels.
 *
 * The reference (aqm857886159/marl) is pure Python and has no FFI of its own; every
 * entry point below therefore cites the Python function whose arithmetic it replaces.  All pointers
 * named *_dev / "device" are device (HBM) pointers owned by the caller; the library owns only the
 * opaque handle (per-environment state records and small tables).  `stream` is a hipStream_t passed
 * as void* (NULL = the default stream).  Every function returns 0 on success and a negative
 * em_status on failure; em_last_error() returns a thread-local message.  No C++ exception crosses
 * this boundary.  A handle must not be used from two host threads at once.
 */
#ifndef EDGEMARL_H
#define EDGEMARL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EM_MAX_NODES 64
#define EM_ABI_VERSION 1

typedef enum em_status {
    EM_OK = 0,
    EM_ERR_INVALID = -1,   /* bad argument (shape, alignment, null pointer, range) */
    EM_ERR_HIP = -2,       /* a HIP runtime call failed; message holds hipGetErrorString */
    EM_ERR_STATE = -3      /* call not valid in the handle's current state */
} em_status;

/* Mirrors ENV_CONFIG (reference MRRL.PY:70-110) and the attributes EdgeSimGym.__init__ copies
 * from it (MRRL.PY:124-186). */
typedef struct em_config {
    int32_t n_nodes;            /* agents == edge servers, 1..EM_MAX_NODES */
    int32_t episode_length;
    int32_t arrival_mode;       /* 0 = midpoint of the range, 1 = "cyclic", 2 = "random" (MRRL.PY:225-235) */
    int32_t disable_env_exploration;
    double node_cpu_capacity[EM_MAX_NODES];
    double arrival_rate_lo, arrival_rate_hi, arrival_cycle_seconds;
    double workload_lo, workload_hi;
    double data_lo, data_hi;
    double deadline_lo, deadline_hi;
    double bw_lo, bw_hi;
    double net_latency_lo, net_latency_hi;
    double w_latency, w_energy, w_violation;
    double min_time_step, max_time_step, initial_observation_window;
    double initial_epsilon, final_epsilon;
    int64_t decay_steps;
} em_config;

/* One step of pre-drawn random numbers for n_envs environments (parity mode).
 * Per-agent arrays are indexed by ACTING SLOT (arrival order), task arrays by arrival index.
 * All pointers are device pointers; unused entries may hold anything. */
typedef struct em_tape {
    const double *explore_u;      /* [n_envs][n_nodes]  np.random.rand()            MRRL.PY:267 */
    const double *explore_res;    /* [n_envs][n_nodes]  uniform(0.1,1.0), float64    MRRL.PY:272 */
    const double *bw_u;           /* [n_envs][n_nodes]  uniform(*bw_range), Mbps     MRRL.PY:412 */
    const double *net_lat;        /* [n_envs][n_nodes]  uniform(*lat_range)          MRRL.PY:413 */
    const int32_t *explore_tgt;   /* [n_envs][n_nodes]  randint(0,n)                 MRRL.PY:271 */
    const double *rate_u;         /* [n_envs]           uniform(lo,hi), "random" mode MRRL.PY:232 */
    const double *dt_raw;         /* [n_envs]           exponential(1/rate), pre-clip MRRL.PY:246 */
    const int32_t *n_arrivals;    /* [n_envs]           poisson(rate*dt)             MRRL.PY:351 */
    const double *task_workload;  /* [n_envs][arrival_slots]                         MRRL.PY:215 */
    const double *task_data;      /* [n_envs][arrival_slots]                         MRRL.PY:216 */
    const double *task_deadline_offset; /* [n_envs][arrival_slots]                   MRRL.PY:217 */
    const int32_t *task_decider;  /* [n_envs][arrival_slots] randint(0,n)            MRRL.PY:355 */
    int32_t arrival_slots;
} em_tape;

typedef struct em_handle em_handle;

/* Outputs of reset/step; any pointer except obs/active may be NULL to skip that output. */
typedef struct em_step_out {
    float *obs;            /* [n_envs][n_nodes][n_nodes+4] f32; zero rows for agents with no task.
                              Row layout = EdgeSimGym._get_obs (MRRL.PY:307-324); the dense tensor is
                              also MAPPOTrainer.obs_dict_to_state (mappo_trainer.py:101-123). */
    uint64_t *active;      /* [n_envs] bit i set <=> agent i has a pending task (key in new obs dict) */
    int8_t *rank;          /* [n_envs][n_nodes] arrival order of the pending task, -1 = none
                              (= insertion order of the reference's obs dict, MRRL.PY:361-362) */
    double *reward;        /* [n_envs] global reward, _calculate_reward (MRRL.PY:326-338) */
    uint8_t *emitted;      /* [n_envs] 1 <=> rewards dict non-empty (MRRL.PY:454-455) */
    uint8_t *done;         /* [n_envs] dones["__all__"] (MRRL.PY:459) */
    double *step_latency;  /* [n_envs][n_nodes] t_total per acting slot, NaN beyond (MRRL.PY:425-428) */
    int32_t *n_draws;      /* [n_envs] random draws the reference would have consumed this call */
    double *reward_seen;   /* [n_envs] reward as the trainer sees it: 0 when not emitted
                              (mappo_trainer.py:209-212) */
} em_step_out;

/* EdgeSimGym.__init__ (MRRL.PY:124-186) for n_envs environments on HIP device `device_id`.
 * seed/env_offset address the Philox stream of the throughput mode: environment b of this handle
 * is global environment env_offset + b (multi-GPU: env_offset = rank * n_envs).
 * latency_log_capacity > 0 keeps that many per-task latencies per env for the exact p99 metric. */
int em_create(const em_config *cfg, int64_t n_envs, int device_id, uint64_t seed, uint64_t env_offset,
              int64_t latency_log_capacity, em_handle **out);
int em_destroy(em_handle *h);

/* EdgeSimGym.reset (MRRL.PY:188-212).  tape == NULL: Philox stream, a new episode index per call;
 * has_seed != 0 re-keys the stream (the reference reseeds numpy's global generator, MRRL.PY:189-191). */
int em_reset(em_handle *h, const em_tape *tape, int has_seed, uint64_t seed, const em_step_out *out, void *stream);

/* EdgeSimGym.step (MRRL.PY:386-521) for every environment of the handle.
 * placement [n_envs][n_nodes] int32 (negative = no action supplied for that agent),
 * resource [n_envs][n_nodes] float32 in [0.1,1] (the dict form of MRRL.PY:280-281). */
int em_step(em_handle *h, const int32_t *placement, const float *resource, const em_tape *tape,
            const em_step_out *out, void *stream);

/* Episode metrics (MRRL.PY:470-504), out6 = device [n_envs][6] float64:
 * avg_latency_ms, p99_latency_ms (NaN without a latency log), avg_energy_J, violation_rate,
 * throughput_tps, load_balance_jain. */
int em_episode_metrics(em_handle *h, double *out6, void *stream);

/* The reference's class-level EdgeSimGym._global_step_counter (MRRL.PY:122,255,467-468) and the
 * disable_env_exploration attribute (mappo_trainer.py:336-346), per handle. */
int em_set_exploration(em_handle *h, int disable_env_exploration, int64_t global_step_counter);
int64_t em_get_exploration_counter(const em_handle *h);

/* Host copies of the per-environment state for tests (synchronises the stream):
 * load/queue [n_envs][n_nodes] f64, sim_time/elapsed [n_envs] f64, step [n_envs] i32,
 * counts [n_envs][3] i64 = tasks, violations, completed. */
int em_get_state(em_handle *h, double *load, double *queue, double *sim_time, double *elapsed, int32_t *step,
                 int64_t *counts, void *stream);

/* Bytes of one environment's state record and of one obs row block (for roofline accounting). */
int64_t em_record_bytes(const em_handle *h);
int64_t em_n_envs(const em_handle *h);
/* Kernel variant: lanes per environment (16/32/64); 0 restores the automatic choice. */
int em_set_lanes_per_env(em_handle *h, int lanes);

int em_abi_version(void);
const char *em_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* EDGEMARL_H */
