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
#define EM_ABI_VERSION 8

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
    uint64_t *chunk_totals; /* [ceil(n_envs/256)] optional: the kernel ADDS, per chunk of 256 environments, the number of
                              agents holding a task after the call (low 32 bits) and of environments with at least one
                              (high 32 bits); zeroed by the caller (or by em_rollout_index_totals).  Not filled by
                              em_step_many.  Feeds em_rollout_index_totals. */
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

/* EdgeSimGym.step fed continuous action vectors (the non-dict branch of _convert_action, MRRL.PY:283-305):
 * raw_actions [n_envs][n_nodes][n_nodes+1] float32; placement = 0 if all of the first n_nodes components are
 * within 1e-8 of 0, else the first argmax; resource = float32(0.1 + 0.9*clip((x_last + 1)/2, 0, 1)) in float64. */
int em_step_raw(em_handle *h, const float *raw_actions, const em_tape *tape, const em_step_out *out, void *stream);

/* EdgeSimGym.step under the greedy baseline's policy (run_greedy_baseline, MRRL.PY:803-812): every acting
 * agent targets argmin(nodes_load / cpu_caps) (first minimum) with the given resource share (reference 0.5). */
int em_step_greedy(em_handle *h, float resource_share, const em_tape *tape, const em_step_out *out, void *stream);

/* k_steps consecutive EdgeSimGym.step calls (MRRL.PY:386-521) in ONE launch, for callers whose actions do not
 * depend on the observations of the steps in between (pre-generated or scripted actions, the greedy baseline): the
 * state records stay on chip between the steps, so a step costs neither a kernel launch nor a record round trip.
 * Bit-identical to k_steps calls of em_step / em_step_raw / em_step_greedy on the in-kernel random stream (no tape).
 * action_mode EM_ACTIONS_DICT: placement [k][n_envs][n] + resource [k][n_envs][n]; EM_ACTIONS_RAW: resource_or_raw =
 * vectors [k][n_envs][n][n+1]; EM_ACTIONS_GREEDY: no arrays, greedy_share.  history bit EM_HISTORY_REWARDS: the
 * reward / reward_seen / emitted / done outputs are [k][n_envs] arrays (one row per step), bit EM_HISTORY_OBS: the same for
 * obs / active / rank; without a bit the output holds the last step only.  No counterpart in the reference (it steps
 * one environment at a time from Python); callers: evaluation baselines and open-loop benchmarks. */
#define EM_ACTIONS_DICT 0
#define EM_ACTIONS_RAW 1
#define EM_ACTIONS_GREEDY 2
#define EM_HISTORY_REWARDS 1
#define EM_HISTORY_OBS 2
int em_step_many(em_handle *h, int32_t k_steps, int32_t action_mode, const int32_t *placement, const float *resource_or_raw,
                 float greedy_share, int32_t history, const em_step_out *out, void *stream);

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
/* Kernel variant: lanes per environment - 16/32/64 (power-of-two groups) or 1 = packed (groups of exactly
 * n_nodes lanes: more environments per wavefront); 0 restores the automatic choice (packed for batches of
 * 65,536 environments or more when it fits more environments per wavefront).  Results do not depend on it. */
int em_set_lanes_per_env(em_handle *h, int lanes);
/* Kernel variant: wavefronts per workgroup of the reset/step kernel (waves - 1 worker wavefronts and one
 * scalar wavefront, 2..8); 0 restores the automatic choice (5 below 65,536 environments, 8 from there on).  Results
 * do not depend on it. */
int em_set_workgroup_waves(em_handle *h, int waves);
/* The variant the next reset/step launch will use (for benchmark and profile labels): lanes a wavefront gives to one
 * environment (n_nodes in the packed layout) and wavefronts per workgroup after the launch's own clamp.  Returns 1 for
 * the packed layout, 0 for power-of-two groups, negative on error.  No counterpart in the reference. */
int em_get_kernel_variant(const em_handle *h, int32_t *lanes_per_env, int32_t *workgroup_waves);


/* ------------------------------------------------------------------------------------------------
 * Scaled topology: D decider agents over S = cfg->n_nodes edge servers.
 *
 * The reference only knows deciders == servers (MRRL.PY:150,153,162,355).  This is the build's generalisation of
 * the same step: decider d is attached to home server d mod S ("local" placement <=> target == home, MRRL.PY:409),
 * arrivals pick a decider uniformly in [0, D) (MRRL.PY:355), a later arrival for a busy decider bumps the queue of its
 * home server (MRRL.PY:358-359), the observation of decider d is _get_obs (MRRL.PY:307-324) seen from its home server
 * (width S + 4), exploration targets are servers (MRRL.PY:271).  With D == S it is the reference step bit for bit
 * (tests/test_scaled_gpu.py compares it with em_step).  Random numbers: the in-kernel Philox stream only.
 *
 * Outputs are COMPACT: at most EMS_PENDING_CAP deciders can hold a task at once (one per arrival of a step), so
 * the library returns the pending list in arrival order instead of a dense [D][S+4] tensor that is >99 % zeros.
 * ---------------------------------------------------------------------------------------------- */
#define EMS_PENDING_CAP 64
#define EMS_INDEX_CURSOR_STRIDE 32
/* chunk totals + two histograms + flag, rounded up to a 128-byte line, then two sets of one line per group cursor (the
 * histograms and cursors alternate by step parity: ems_rollout_index clears the set of step t + 1 while it uses that of t) */
#define EMS_INDEX_SCRATCH_INTS(n_envs, n_deciders) \
    (((((n_envs) + 255) / 256 + 2 * (n_deciders) + 1 + 31) / 32) * 32 + 64 * (int64_t)(n_deciders))
#define EMS_MAX_DECIDERS 4096

typedef struct ems_handle ems_handle;

typedef struct ems_step_out {
    int32_t *n_pending;     /* [n_envs] deciders holding a task after the call (keys of the new obs dict)     */
    int32_t *pend_decider;  /* [n_envs][EMS_PENDING_CAP] decider ids in arrival order (obs dict insertion order) */
    float *pend_obs;        /* [n_envs][EMS_PENDING_CAP][S+4] observation rows of those deciders (_get_obs)      */
    double *reward;         /* [n_envs] _calculate_reward (MRRL.PY:326-338); may be NULL                        */
    double *reward_seen;    /* [n_envs] 0 when the new obs set is empty (mappo_trainer.py:209-212); may be NULL  */
    uint8_t *emitted;       /* [n_envs] rewards dict non-empty (MRRL.PY:454-455); may be NULL                   */
    uint8_t *done;          /* [n_envs] dones["__all__"] (MRRL.PY:459); may be NULL                             */
    double *step_latency;   /* [n_envs][EMS_PENDING_CAP] t_total per acting slot, NaN beyond; may be NULL         */
} ems_step_out;

/* EdgeSimGym.__init__ (MRRL.PY:124-186) with n_deciders agents over cfg->n_nodes servers. */
int ems_create(const em_config *cfg, int32_t n_deciders, int64_t n_envs, int device_id, uint64_t seed,
               uint64_t env_offset, int64_t latency_log_capacity, ems_handle **out);
int ems_destroy(ems_handle *h);
/* EdgeSimGym.reset (MRRL.PY:188-212). */
int ems_reset(ems_handle *h, int has_seed, uint64_t seed, const ems_step_out *out, void *stream);
/* EdgeSimGym.step (MRRL.PY:386-521).  actions_by_slot == 0: placement/resource are [n_envs][n_deciders] (indexed by
 * decider, the dict form of MRRL.PY:280-281); != 0: [n_envs][EMS_PENDING_CAP] aligned with the pending list the
 * previous call returned.  placement = target server, negative = no action for that decider. */
int ems_step(ems_handle *h, const int32_t *placement, const float *resource, int actions_by_slot,
             const ems_step_out *out, void *stream);
/* Episode metrics (MRRL.PY:470-504), out6 as em_episode_metrics. */
int ems_episode_metrics(ems_handle *h, double *out6, void *stream);
int ems_set_exploration(ems_handle *h, int disable_env_exploration, int64_t global_step_counter);
int64_t ems_get_exploration_counter(const ems_handle *h);
/* Host copies for tests: load/queue [n_envs][S] f64, sim_time/elapsed [n_envs], step [n_envs], counts [n_envs][3]. */
int ems_get_state(ems_handle *h, double *load, double *queue, double *sim_time, double *elapsed, int32_t *step,
                  int64_t *counts, void *stream);
int64_t ems_record_bytes(const ems_handle *h);
/* Kernel variant: environments per wavefront of the reset/step kernel: 4 (16-lane groups for the pending slots and
 * arrivals; automatic from 16,384 environments) or 1; 0 restores the automatic choice.  Results do not depend on it. */
int ems_set_envs_per_wave(ems_handle *h, int envs_per_wave);

/* Device buffers of one rollout on the scaled topology (RolloutBuffer, mappo_algorithm.py:93-131; all caller-owned).
 * Entries = acting (environment, decider) transitions in (step, environment, arrival slot) order. */
typedef struct ems_rollout {
    int64_t n_envs, capacity_entries;
    int32_t n_servers, n_deciders, horizon, pad;
    float *ent_obs;        /* [E][n_servers+4]                                                        */
    float *ent_action;     /* [E][n_servers+1] raw action vector                                      */
    float *ent_logp;       /* [E]                                                                     */
    float *ent_value;      /* [E] critic value of the entry's observation (independent learners)      */
    int32_t *ent_env;      /* [E]                                                                     */
    int32_t *ent_agent;    /* [E] decider id                                                          */
    int32_t *ent_dst;      /* [E] env * EMS_PENDING_CAP + slot: where the entry's env action goes     */
    float *ent_adv;        /* [E]                                                                     */
    float *ent_ret;        /* [E]                                                                     */
    double *reward_tb;     /* [T][B] ems_step_out.reward_seen                                         */
    uint8_t *done_tb;      /* [T][B]                                                                  */
    uint8_t *k_tb;         /* [T][B] entries of the step                                              */
    int32_t *first_tb;     /* [T][B] first entry of the step                                          */
    int64_t *counters;     /* [T+1]  entry base of each step; element 0 zeroed by the caller          */
    int64_t *group_off;    /* [n_deciders+3] of the CURRENT step: positions of each group in perm, then
                              the pair {0, entries of the step} (the step as one group)               */
    int32_t *perm;         /* [n_envs*EMS_PENDING_CAP] of the CURRENT step: position -> entry index   */
    int32_t *err_flag;     /* [1] sticky error bits (8 = capacity overflow), zeroed by the caller     */
    int32_t *ent_state;    /* [E] state row of the entry = step * n_envs + env (central critic); may be NULL */
    float *value_tb;       /* [T][B] central-critic value of the (step, env) state; may be NULL        */
    int32_t *index_scratch; /* [EMS_INDEX_SCRATCH_INTS(B, n_deciders)] scratch of ems_rollout_index, 128-byte aligned,
                               zeroed by the caller once */
} ems_rollout;

/* Records which deciders act at step t (the pending lists ems_reset/ems_step returned): numbers the entries, copies
 * their observation rows, and groups the step's entries by serving network (n_groups = n_deciders: one network per
 * decider; 1: a shared network) for the grouped MLP kernels below.  Steps are indexed in order (t = 0, 1, ...: step t
 * reads counters[t] and writes counters[t + 1]); pend_decider must be 16-byte aligned (EM_ERR_INVALID otherwise).  Two
 * launches per call; EMS_INDEX_THREE_LAUNCHES=1 in the environment selects the older three-launch path (same records).
 * pend_obs NULL: the observation rows are NOT copied here (the caller's em_gmlp_policy_gather launch of the step copies
 * them while it reads them). */
int ems_rollout_index(const ems_rollout *ro, int32_t t, int32_t n_groups, const int32_t *n_pending,
                      const int32_t *pend_decider, const float *pend_obs, void *stream);

/* ------------------------------------------------------------------------------------------------
 * MAPPO side (reference mappo_algorithm.py / mappo_trainer.py)
 * ---------------------------------------------------------------------------------------------- */

/* Parameters of a 3-layer ReLU MLP in torch.nn.Linear layout (weight [out][in] row-major, float32):
 * MAPPOActor fc1/fc2/mu (mappo_algorithm.py:28-41) or MAPPOCritic fc1/fc2/value_out (:76-90). */
typedef struct em_mlp3 {
    const float *w1, *b1;   /* [hidden][in_dim], [hidden] */
    const float *w2, *b2;   /* [hidden][hidden], [hidden] */
    const float *w3, *b3;   /* [out_dim][hidden], [out_dim] */
    int32_t in_dim, hidden, out_dim;   /* hidden must be 64; in_dim <= 144; out_dim <= 16 */
} em_mlp3;

/* Device buffers of one rollout of `horizon` steps over n_envs environments; everything is
 * allocated by the caller.  "entries" are acting (env, step, agent) transitions in (step, env,
 * arrival-rank) order = RolloutBuffer rows (mappo_algorithm.py:93-131); "steps" are the (step, env)
 * pairs with at least one acting agent, whose dense state row is the critic input
 * (mappo_trainer.py:101-123).  The *_tb arrays are dense [horizon][n_envs]. */
typedef struct em_rollout {
    int64_t n_envs, capacity_entries, capacity_steps;
    int32_t n_nodes, horizon;
    float *ent_obs;        /* [E][n_nodes+4]  observation of the acting agent                     */
    float *ent_action;     /* [E][n_nodes+1]  raw action vector (pre argmax/tanh)                  */
    float *ent_logp;       /* [E]                                                                 */
    int32_t *ent_env;      /* [E]                                                                 */
    int8_t *ent_agent;     /* [E]                                                                 */
    float *ent_adv;        /* [E]  GAE advantage (un-normalised), written by em_gae               */
    float *ent_ret;        /* [E]  return = advantage + value                                     */
    float *st_state;       /* [S][n_nodes*(n_nodes+4)]                                            */
    float *st_value;       /* [S]  critic value of the pre-step state, de-normalised             */
    int32_t *st_env;       /* [S]                                                                 */
    int32_t *st_first;     /* [S]  first entry of this step                                       */
    uint8_t *st_k;         /* [S]  number of entries of this step                                 */
    double *reward_tb;     /* [T][B] reward as the trainer sees it (em_step_out.reward_seen)      */
    uint8_t *done_tb;      /* [T][B]                                                              */
    float *value_tb;       /* [T][B] value of steps with acting agents                            */
    uint8_t *k_tb;         /* [T][B] acting agents                                                */
    int32_t *first_tb;     /* [T][B] first entry                                                  */
    int32_t *srow;         /* [B]    scratch: step row of the current step or -1                  */
    int64_t *scan_scratch; /* [ceil(B/256)] scratch of em_rollout_index                           */
    int64_t *counters;     /* [T+1][4] {entry_base, n_entries, step_base, n_steps}; row 0 zeroed by the caller */
    int32_t *err_flag;     /* [1] sticky error bits (8 = capacity overflow), zeroed by the caller */
    /* States BY REFERENCE (st_state == NULL): the rollout keeps the dense observation tensors of its steps,
     * obs_tb [slices][n_envs][n_nodes][n_nodes+4] (the step kernel writes each step's observations straight into the
     * next slice), and a state is the row number of its environment's block there; nothing but the acting agents' rows
     * is copied when a step is recorded.  With st_state != NULL the state rows are copied (compact) and these two
     * fields are ignored. */
    const float *obs_tb;   /* base of the observation history, or NULL                                   */
    int32_t *st_row;       /* [S]  slice * n_envs + env of the state's block in obs_tb                    */
} em_rollout;

/* y[rows][out_dim] = MLP(x[rows][ldx]) on the f32 matrix cores (MAPPOActor.forward / MAPPOCritic.forward). */
int em_mlp3_forward(const em_mlp3 *net, const float *x, int64_t ldx, int64_t n_rows, float *y, void *stream);

/* MAPPOActor.get_action_log_prob given actions (mappo_algorithm.py:50-71): per row, log N(action; mu, exp(log_std)) summed
 * over the out_dim components and the entropy of that Gaussian summed likewise.  mu/action [n_rows][out_dim], log_std
 * [out_dim]; either output may be NULL. */
int em_gauss_logp(const float *mu, const float *log_std, const float *action, int64_t n_rows, int32_t out_dim,
                  float *logp_out, float *entropy_out, void *stream);
/* Normal(mu, exp(log_std)).rsample() on pre-drawn standard normals eps [n_rows][out_dim] (mappo_algorithm.py:62-65):
 * action = mu + eps * exp(log_std). */
int em_gauss_sample(const float *mu, const float *log_std, const float *eps, int64_t n_rows, int32_t out_dim,
                    float *action_out, void *stream);
/* MAPPOTrainer._format_env_actions (mappo_trainer.py:125-158) per row of raw action vectors [n_rows][n_nodes+1]:
 * placement = 0 when the first n_nodes components are all within 1e-8 of 0, else their first argmax; resource =
 * float32(0.1 + 0.9 * clip((tanh(x_last) + 1) / 2, 0, 1)), tanh in float64. */
int em_format_actions(const float *raw_actions, int64_t n_rows, int32_t n_nodes, int32_t *placement_out,
                      float *resource_out, void *stream);

/* Record which agents act at step t: indexes and compacts the observations of the environment's
 * current obs/active/rank outputs into the rollout buffers (RolloutBuffer.add of obs and state,
 * mappo_trainer.py:160-175). */
int em_rollout_index(const em_rollout *ro, int32_t t, const uint64_t *active, const int8_t *rank, const float *obs,
                     void *stream);

/* em_rollout_index for a caller that already holds the per-256-environment totals of the step's acting agents
 * (em_step_out.chunk_totals: the reset/step kernel counts them while it writes the observations), which saves the
 * counting launch.  chunk_totals [ceil(n_envs/256)]: low 32 bits = acting agents, high 32 bits = environments with at
 * least one; chunk_totals_next (optional) is cleared for the next step's kernel to add into (two buffers used
 * alternately).  No counterpart in the reference. */
int em_rollout_index_totals(const em_rollout *ro, int32_t t, const uint64_t *active, const int8_t *rank, const float *obs,
                            const uint64_t *chunk_totals, uint64_t *chunk_totals_next, void *stream);

/* em_rollout_index_totals and the actor half of em_policy_step in ONE launch (RolloutBuffer.add + _store_transition +
 * MAPPO.get_actions + _format_env_actions, mappo_algorithm.py:104-111,189-221, mappo_trainer.py:125-175): each workgroup
 * numbers and copies the entries of its 256 environments and runs the actor on exactly those entries, reading their
 * observations from `obs` directly.  Results are bit-identical to the two separate calls.  Needs the states held by
 * reference (st_state == NULL, obs_tb, st_row), chunk_totals from the step kernel and n_nodes <= 12; the critic is not
 * evaluated (em_value_all or em_value_step do that).  Arguments as in the two calls it replaces. */
int em_rollout_policy_step(const em_rollout *ro, int32_t t, const uint64_t *active, const int8_t *rank, const float *obs,
                           const uint64_t *chunk_totals, uint64_t *chunk_totals_next, const em_mlp3 *actor,
                           const float *log_std, const float *noise, int explore, uint64_t seed, uint32_t episode,
                           uint32_t rng_step, uint64_t env_offset, int32_t *placement, float *resource, void *stream);

/* MAPPO.get_actions + MAPPOTrainer._format_env_actions + MAPPO.get_values for step t
 * (mappo_algorithm.py:189-239, mappo_trainer.py:125-158): samples a = mu + sigma*eps for every entry
 * of the step (noise = pre-drawn eps [E][n+1] aligned with the entries, or NULL for the in-kernel
 * Philox/Box-Muller stream addressed by (seed; env_offset+env, episode, rng_step, agent); explore == 0 takes
 * a = mu and log-prob 0), stores action/log-prob, writes
 * the env actions placement/resource [n_envs][n_nodes] and the critic values.  critic == NULL skips the value part
 * (a caller that runs em_value_step itself, e.g. on a second stream beside the environment step). */
int em_policy_step(const em_rollout *ro, int32_t t, const em_mlp3 *actor, const float *log_std, const em_mlp3 *critic,
                   float value_norm_mean, float value_norm_var, const float *noise, int explore, uint64_t seed,
                   uint32_t episode, uint32_t rng_step, uint64_t env_offset, int32_t *placement, float *resource,
                   void *stream);
/* MAPPO.get_values for step t alone (mappo_algorithm.py:226-239): the critic forward over the step's compact states,
 * results into st_value and value_tb[t].  Needs em_rollout_index of step t; nothing before em_gae reads its output. */
int em_value_step(const em_rollout *ro, int32_t t, const em_mlp3 *critic, float value_norm_mean, float value_norm_var,
                  void *stream);

/* MAPPO.get_values (mappo_algorithm.py:226-239) for ALL states of a rollout that holds them by reference, in one launch:
 * the parameters do not change during a rollout and nothing before em_gae reads a value, so the critic need not run once
 * per step (a thousand latency-bound launches) - the same rows through the same kernel give the same bits.  Slice i of
 * obs_tb must hold the observations of recorded step i (value_tb[st_row[s]] receives the value of state s). */
int em_value_all(const em_rollout *ro, int64_t n_states, const em_mlp3 *critic, float value_norm_mean, float value_norm_var,
                 void *stream);

/* MAPPO.compute_gae (mappo_algorithm.py:241-262) for every environment's segment of the transition
 * list (next_value = 0 per segment, :284), float64 recursion, results rounded to float32 into
 * ro->ent_adv / ro->ent_ret.  moments3 (device, 3 doubles) receives {count, sum, sum of squares} of the
 * float32 advantages for the global normalisation (:293); partials is scratch of
 * 3*ceil(n_envs/256) doubles. */
int em_gae(const em_rollout *ro, double gamma, double lam, double *partials, int64_t partials_capacity,
           double *moments3, void *stream);

/* em_gae for callers that cannot allocate or carry a workspace (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: same arithmetic and outputs as em_gae
 * (mappo_algorithm.py:241-262,293), but the per-workgroup moment partials live in a fixed array inside the library
 * (96 KB per GPU, held until the process ends) instead of a caller-owned buffer.  The call allocates nothing (no
 * hipMalloc/hipFree, no allocating library routine); its first call on a device resolves the array's device address
 * (hipGetSymbolAddress), so make one call per device before a graph capture begins.  n_envs above
 * EM_GAE_STATIC_MAX_ENVS is rejected with EM_ERR_INVALID.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (two streams, two host threads) race; calls on one stream are ordered and need no clearing.
 * The array used is the CURRENT device's: `stream` (and the rollout buffers) must belong to the current device; a stream of
 * another device is rejected with EM_ERR_INVALID. */
#define EM_GAE_STATIC_MAX_ENVS (1 << 20)
int em_gae_static(const em_rollout *ro, double gamma, double lam, double *moments3, void *stream);

/* Floats per slab row used by the two gradient kernels below: all parameters of the network in the
 * order [w1|b1|w2|b2|w3|b3|(log_std)] followed by 3 loss sums. */
int64_t em_ppo_slab_floats(int32_t in_dim, int32_t out_dim, int is_actor);

/* One epoch of MAPPO.update for the shared actor (mappo_algorithm.py:302,306-309,317-320): recomputes
 * mu(obs), the PPO-clip objective with the globally normalised advantages and the entropy bonus, and
 * back-propagates on the matrix cores.  grad_out[ n_params ] = d(actor_loss + entropy_coeff * (-entropy))
 * / d params for THIS rank's entries with the mean taken over n_global transitions (sum the outputs of
 * all ranks for data parallelism).  loss3_out = {sum min(surr1,surr2), sum entropy, entries}.
 * slab = scratch [n_waves][em_ppo_slab_floats]; n_waves = 4 * workgroups to launch. */
int em_ppo_actor_grad(const em_mlp3 *actor, const float *log_std, const float *ent_obs, const float *ent_action,
                      const float *ent_logp, const float *ent_adv, const double *adv_moments3, int64_t n_entries,
                      int64_t n_global, float clip, float entropy_coeff, float *slab, int64_t n_waves,
                      float *grad_out, float *loss3_out, void *stream);

/* Same for the central critic (mappo_algorithm.py:303,312-314): value_loss_coeff * 0.5 * mean(e^2) with
 * e = normalise(v) - normalise(return); the critic is evaluated once per (env, step) and weighted by its
 * k transitions, which equals the reference's per-transition evaluation of identical states.
 * loss3_out = {sum e^2, 0, transitions}. */
int em_ppo_critic_grad(const em_mlp3 *critic, const float *st_state, const int32_t *st_first, const uint8_t *st_k,
                       const float *ent_ret, int64_t n_steps, int64_t n_global, float value_loss_coeff,
                       float value_norm_mean, float value_norm_var, float *slab, int64_t n_waves, float *grad_out,
                       float *loss3_out, void *stream);

/* em_ppo_critic_grad with the states held BY REFERENCE (em_rollout.obs_tb / st_row): state s is the n_nodes*(n_nodes+4)
 * floats at state_base + st_row[s] * in_dim; st_row == NULL reads state s at row s (= em_ppo_critic_grad). */
int em_ppo_critic_grad_rows(const em_mlp3 *critic, const float *state_base, const int32_t *st_row, const int32_t *st_first,
                            const uint8_t *st_k, const float *ent_ret, int64_t n_steps, int64_t n_global,
                            float value_loss_coeff, float value_norm_mean, float value_norm_var, float *slab,
                            int64_t n_waves, float *grad_out, float *loss3_out, void *stream);

/* clip_grad_norm_ on each network separately (mappo_algorithm.py:328-329) then torch.optim.Adam's
 * update (:331-332) on flat float32 parameter/moment buffers; step counts from 1.
 * norms2_out (optional) receives the two pre-clip gradient norms. */
int em_adam_step(float *actor_param, const float *actor_grad, float *actor_m, float *actor_v, int64_t n_actor,
                 double lr_actor, float *critic_param, const float *critic_grad, float *critic_m, float *critic_v,
                 int64_t n_critic, double lr_critic, int64_t step, double beta1, double beta2, double eps,
                 double max_grad_norm, float *norms2_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Grouped MLP kernels for the scaled topology (BASELINE.json configs[4]): IPPO = one actor and one local critic per
 * decider agent, MAPPO = one shared actor (n_groups == 1).  Build-defined: the reference's IPPO is RLlib's PPO with
 * one policy per agent (MRRL.PY:611-638, un-vendored ray[rllib]); here every independent learner runs the reference's
 * own MAPPO arithmetic (mappo_algorithm.py) on its own transitions.  Rows are served in GROUPS: positions
 * [group_off[g], group_off[g+1]) of the grouped order use network g; perm maps a position to its row in x and in the
 * per-row arrays (NULL = identity; group_off NULL needs n_groups == 1).  split = workgroups per group.
 * ---------------------------------------------------------------------------------------------- */
typedef struct em_gmlp {
    const float *w1, *b1, *w2, *b2, *w3, *b3;   /* network 0, torch.nn.Linear layout as em_mlp3 */
    int64_t group_stride;                       /* floats between the same parameter of consecutive networks */
    int32_t in_dim, hidden, out_dim, n_groups;  /* hidden 64, in_dim <= 80, out_dim <= 80 */
} em_gmlp;

/* y[row][out_dim] = MLP_g(x[row]) (MAPPOActor.forward / MAPPOCritic.forward, mappo_algorithm.py:36-41,83-90). */
int em_gmlp_forward(const em_gmlp *net, const float *x, int64_t ldx, int64_t n_rows, const int64_t *group_off,
                    const int32_t *perm, int32_t split, float *y, void *stream);
/* MAPPO.get_actions + MAPPOTrainer._format_env_actions per row (mappo_algorithm.py:189-224, mappo_trainer.py:125-158):
 * a = mu + exp(log_std) * eps (noise [rows][out_dim] pre-drawn, or NULL for the in-kernel Philox/Box-Muller stream
 * addressed by (seed; env_offset + ent_env, episode, rng_step, ent_agent)), log-prob, placement = first argmax of the
 * first n_place components (0 if all within 1e-8 of 0), resource = float32(0.1 + 0.9 clip((tanh(a[n_place]) + 1)/2)),
 * written to placement/resource[ent_dst[row]].  log_std: network 0's, same group stride. */
int em_gmlp_policy(const em_gmlp *actor, const float *log_std, const float *x, int64_t ldx, int64_t n_rows,
                   const int64_t *group_off, const int32_t *perm, int32_t split, const float *noise, int explore,
                   uint64_t seed, uint32_t episode, uint32_t rng_step, uint64_t env_offset, const int32_t *ent_env,
                   const int32_t *ent_agent, const int32_t *ent_dst, int32_t n_place, float *ent_action, float *ent_logp,
                   int32_t *placement, float *resource, void *stream);
/* em_gmlp_policy for rows whose observations still lie in the environment's pending lists (ems_step_out.pend_obs viewed as
 * [n_envs * EMS_PENDING_CAP][ldx]): row r reads x_src[ent_dst[r]] and stores it to x_copy[r] ([rows][ldx], the entry storage
 * ems_rollout.ent_obs) on its way - the observation copy of RolloutBuffer.add (mappo_algorithm.py:104-111) that
 * ems_rollout_index leaves out when its pend_obs argument is NULL.  Everything else as em_gmlp_policy. */
int em_gmlp_policy_gather(const em_gmlp *actor, const float *log_std, const float *x_src, float *x_copy, int64_t ldx,
                          int64_t n_rows, const int64_t *group_off, const int32_t *perm, int32_t split, const float *noise,
                          int explore, uint64_t seed, uint32_t episode, uint32_t rng_step, uint64_t env_offset,
                          const int32_t *ent_env, const int32_t *ent_agent, const int32_t *ent_dst, int32_t n_place,
                          float *ent_action, float *ent_logp, int32_t *placement, float *resource, void *stream);
/* MAPPO.get_values per row (mappo_algorithm.py:226-239): value[row] = critic_g(x[row]) de-normalised. */
int em_gmlp_value(const em_gmlp *critic, const float *x, int64_t ldx, int64_t n_rows, const int64_t *group_off,
                  const int32_t *perm, int32_t split, float value_norm_mean, float value_norm_var, float *value,
                  void *stream);
/* Floats per slab row / per network gradient (+3 loss sums): [w1|b1|w2|b2|w3|b3|(log_std)|3]. */
int64_t em_gmlp_slab_floats(int32_t in_dim, int32_t out_dim, int is_actor);
/* One epoch of MAPPO.update per group (mappo_algorithm.py:293,302-309,317-320): advantages normalised with the
 * group's moments3[g] = {count, sum, sum of squares} (global over ranks; fewer than 2 transitions: variance 0), means
 * over the group's count.  grad_out [n_groups][slab_floats - 3], loss3_out [n_groups][3] = {sum min(surr), sum
 * entropy, rows}; slab = scratch [n_groups][4 * split][slab_floats]. */
int em_gmlp_actor_grad(const em_gmlp *actor, const float *log_std, const float *x, int64_t ldx, const float *action,
                       const float *old_logp, const float *adv, const double *moments3, int64_t n_rows,
                       const int64_t *group_off, const int32_t *perm, int32_t split, float clip, float entropy_coeff,
                       float *slab, float *grad_out, float *loss3_out, void *stream);
/* Value loss per group (mappo_algorithm.py:303,312-314) on per-row returns; loss3_out = {sum e^2, 0, rows}. */
int em_gmlp_critic_grad(const em_gmlp *critic, const float *x, int64_t ldx, const float *ret, const double *moments3,
                        int64_t n_rows, const int64_t *group_off, const int32_t *perm, int32_t split,
                        float value_loss_coeff, float value_norm_mean, float value_norm_var, float *slab, float *grad_out,
                        float *loss3_out, void *stream);
/* clip_grad_norm_ per network (mappo_algorithm.py:328-329) + Adam (:331-332) on [n_groups][n_per_group] flat buffers. */
int em_group_adam(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n_per_group,
                  int64_t n_groups, double lr, int64_t step, double beta1, double beta2, double eps, double max_grad_norm,
                  float *norms_out, void *stream);
/* moments3[g] = {rows, sum, sum of squares} of adv over group g, fixed summation order. */
int em_group_moments(const float *adv, const int64_t *group_off, const int32_t *perm, int32_t n_groups, double *moments3,
                     void *stream);
/* MAPPO.compute_gae (mappo_algorithm.py:241-262) for independent learners: one chain per (environment, agent) over the
 * agent's own transitions, next_value = 0 at the end of the recorded horizon.  reward/done/k/first are dense
 * [horizon][n_envs] (k = entries of the step, first = index of its first entry), ent_agent/ent_value per entry;
 * chain_state = scratch [n_envs][n_agents][2] doubles. */
int em_gae_agents(int64_t n_envs, int32_t horizon, int32_t n_agents, const double *reward_tb, const uint8_t *done_tb,
                  const uint8_t *k_tb, const int32_t *first_tb, const int32_t *ent_agent, const float *ent_value,
                  double gamma, double lam, double *chain_state, float *ent_adv, float *ent_ret, void *stream);

/* Groups n_rows rows by key in [0, n_groups) (entry -> the learner that owns it: the reference trains one PPO learner per
 * agent, MRRL.PY:611-638) for the grouped kernels above: group_off [n_groups + 1] (int64) and perm [n_rows] (position ->
 * row), STABLE (rows of a group keep their order, so every learner sums in the same order from run to run).  A counting sort
 * in four short launches; scratch: em_group_index_scratch_ints(n_rows, n_groups) int32, need not be zeroed.  n_groups <= 1024,
 * n_rows < 2^31; keys outside the range are clamped into it. */
int64_t em_group_index_scratch_ints(int64_t n_rows, int32_t n_groups);
int em_group_index(const int32_t *key, int64_t n_rows, int32_t n_groups, int64_t *group_off, int32_t *perm, int32_t *scratch,
                   void *stream);

/* MAPPO's central critic on the scaled topology (MAPPOCritic, mappo_algorithm.py:74-90, on the concatenated state of
 * mappo_trainer.py:101-123): D*width -> 64 -> 64 -> 1 with the first layer evaluated sparsely over the deciders that
 * hold a task (absent deciders are zero rows of the state).  Parameters in torch.nn.Linear layout; w1 is [64][ldw],
 * decider d owns columns [d*width, (d+1)*width). */
typedef struct em_central_critic {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    int32_t n_deciders, width, ldw, pad;
} em_central_critic;

/* MAPPO.get_values (mappo_algorithm.py:226-239) for n_states states: state s has k_arr[s] entries starting at entry
 * first_arr[s]; group_off/perm group the entries by decider (ems_rollout_index with n_groups = n_deciders).
 * partial = scratch [entries][64]; value_out[s] is written for states with at least one entry. */
int em_cc_forward(const em_central_critic *net, const float *ent_obs, const int64_t *group_off, const int32_t *perm,
                  int32_t split, float *partial, int64_t n_states, const uint8_t *k_arr, const int32_t *first_arr,
                  float value_norm_mean, float value_norm_var, float *value_out, void *stream);
/* Floats of the `slab` scratch em_cc_grad needs. */
int64_t em_cc_scratch_floats(const em_central_critic *net, int32_t split, int64_t n_states);
/* One epoch of the critic part of MAPPO.update (mappo_algorithm.py:303,312-314): value_loss_coeff * 0.5 * mean over the
 * n_global transitions of (norm(v) - norm(return))^2, every state evaluated once and weighted by its transitions.
 * grad_out = [w1 (64 x ldw) | b1 | w2 | b2 | w3 | b3], loss3_out = {sum e^2, 0, transitions}; dz = scratch
 * [n_states][64], partial = scratch [entries][64]. */
int em_cc_grad(const em_central_critic *net, const float *ent_obs, const int32_t *ent_state, const float *ent_ret,
               const int64_t *group_off, const int32_t *perm, int32_t split, float *partial, int64_t n_states,
               const uint8_t *k_arr, const int32_t *first_arr, int64_t n_global, float value_loss_coeff,
               float value_norm_mean, float value_norm_var, float *dz, float *slab, float *grad_out, float *loss3_out,
               void *stream);

/* ------------------------------------------------------------------------------------------------
 * EdgeMARLEnv: the PyMARL twin environment (pymarl/src/envs/edge_marl_env.py), batched.  Its dynamics are not
 * EdgeSimGym's: every agent holds a task every step, the clock advances before the placements, a placement adds the
 * task's workload to the target's load, the reward uses the violation rate.  SURVEY.md 8f rank 4.
 * ------------------------------------------------------------------------------------------------ */

/* EdgeMARLEnv.__init__ (edge_marl_env.py:18-58), defaults of config/envs/edge_marl.yaml. */
typedef struct emm_config {
    int32_t n_nodes;            /* agents == edge servers, 1..EM_MAX_NODES */
    int32_t episode_length;     /* episode_limit */
    int32_t arrival_mode;       /* 0 = fixed task_arrival_rate, 1 = "cyclic", 2 = "random" (edge_marl_env.py:236-243) */
    int32_t action_mode;        /* 0 = "discrete", 1 = "hybrid" (edge_marl_env.py:98-104) */
    double node_cpu_capacity[EM_MAX_NODES];
    double arrival_rate, arrival_rate_lo, arrival_rate_hi, arrival_cycle_seconds;
    double workload_lo, workload_hi;
    double data_lo, data_hi;
    double deadline_lo, deadline_hi;
    double bw_lo, bw_hi;
    double net_latency_lo, net_latency_hi;
    double alpha, beta, gamma;  /* reward_weights */
    double min_time_step, max_time_step;
} emm_config;

/* One step of pre-drawn random numbers (parity mode).  Device pointers; entries the reference would not have drawn
 * (local placements, rate_u outside "random" mode) may hold anything. */
typedef struct emm_tape {
    const double *rate_u;               /* [n_envs]           rng.uniform(*rate_range)      edge_marl_env.py:242 */
    const double *dt_raw;               /* [n_envs]           rng.exponential(1/rate)       :233 */
    const double *bw_u;                 /* [n_envs][n_nodes]  rng.uniform(*bw_range), Mbps  :111 */
    const double *net_lat;              /* [n_envs][n_nodes]  rng.uniform(*latency_range)   :112 */
    const double *task_workload;        /* [n_envs][n_nodes]  :246 */
    const double *task_data;            /* [n_envs][n_nodes]  :247 */
    const double *task_deadline_offset; /* [n_envs][n_nodes]  :248 */
} emm_tape;

/* Outputs of reset/step; any pointer may be NULL to skip that output (reset fills obs only). */
typedef struct emm_step_out {
    float *obs;           /* [n_envs][n_nodes][n_nodes+4] f32 = get_obs (edge_marl_env.py:164-165,265-288); the same
                             buffer read as [n_envs][n_nodes*(n_nodes+4)] is get_state (:173-174) */
    double *reward;       /* [n_envs] (:145,290-294) */
    uint8_t *terminated;  /* [n_envs] (:146) == info["episode_limit"] (:160) */
    double *info;         /* [n_envs][6]: avg_latency_ms, avg_energy_J, deadline_violation_rate, throughput_tps,
                             load_balance_jain (:154-160), then the mean latency in seconds as episode_metrics keeps it
                             (:148; get_episode_summary, :193-207, is computed from a log of these rows) */
    double *latency;      /* [n_envs][n_nodes] total_latency of each agent's task (:119) */
} emm_step_out;

typedef struct emm_handle emm_handle;

/* seed/env_offset address the Philox stream of the throughput mode (environment b = global env_offset + b). */
int emm_create(const emm_config *cfg, int64_t n_envs, int device_id, uint64_t seed, uint64_t env_offset, emm_handle **out);
int emm_destroy(emm_handle *h);
/* EdgeMARLEnv.reset (edge_marl_env.py:67-77); tape: only the three task arrays are read.  has_seed re-keys the
 * Philox stream (EdgeMARLEnv.seed, :60-62). */
int emm_reset(emm_handle *h, const emm_tape *tape, int has_seed, uint64_t seed, const emm_step_out *out, void *stream);
/* EdgeMARLEnv.step (edge_marl_env.py:79-162).  Discrete mode: actions_i32 [n_envs][n_nodes], clipped to [0, n-1];
 * hybrid mode: actions_f32 [n_envs][n_nodes][n_nodes+1] (_convert_hybrid_action, :304-318). */
int emm_step(emm_handle *h, const int32_t *actions_i32, const float *actions_f32, const emm_tape *tape,
             const emm_step_out *out, void *stream);
/* Kernel variant: 1 lane per environment (large batches) or 16 lanes per environment (n_nodes <= 16; small batches);
 * 0 = the library's choice (16 lanes up to 32,768 environments).  Identical results; tuning and tests only.  No
 * counterpart in the reference. */
int emm_set_lanes_per_env(emm_handle *h, int32_t lanes_per_env);
/* Raw state for tests: load [n_envs][n] f64, queue [n_envs][n] f64, sim_time [n_envs] f64, step [n_envs] i32. */
int emm_get_state(emm_handle *h, double *load, double *queue, double *sim_time, int32_t *step, void *stream);

int em_abi_version(void);
const char *em_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* EDGEMARL_H */
