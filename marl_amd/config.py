"""Environment configuration: the reference's ENV_CONFIG keys (MRRL.PY:70-110), unchanged."""
from __future__ import annotations

import ctypes as C

import numpy as np

EM_MAX_NODES = 64

ENV_CONFIG = {
    "n_nodes": 10,
    "task_arrival_rate": 10,
    "task_arrival_rate_range": (5, 15),
    "task_arrival_mode": "cyclic",
    "task_arrival_cycle_seconds": 20.0,
    "episode_length": 1000,
    "node_cpu_capacity": np.array([1.0, 1.2, 1.4, 1.6, 1.8, 2.0, 2.2, 2.4, 2.6, 3.0]) * 1e9,
    "task_workload_range": (1.0, 10.0),
    "task_data_range": (0.5, 5.0),
    "task_deadline_range": (0.05, 0.5),
    "network_bw_range": (10, 100),
    "network_latency_range": (0.002, 0.01),
    "reward_weights": {"alpha": 0.5, "beta": 0.3, "gamma": 0.2},
    "min_time_step": 0.02,
    "max_time_step": 0.2,
    "initial_observation_window": 0.1,
    "exploration_schedule": {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 1_000_000},
    "disable_env_exploration": False,
}

# Scaled topology of BASELINE.json configs[4] / SURVEY.md 8d config 5 (build-defined generalisation, not in the
# reference): 64 edge servers, 256 decider agents (home server d mod 64), 10x the arrival rate.
SCALED_ENV_CONFIG = dict(ENV_CONFIG)
SCALED_ENV_CONFIG.update({
    "n_nodes": 64,
    "n_deciders": 256,
    "node_cpu_capacity": np.linspace(1.0, 3.0, 64) * 1e9,
    "task_arrival_rate": 100,
    "task_arrival_rate_range": (50, 150),
})


class EmConfig(C.Structure):
    """ctypes image of ``em_config`` (include/edgemarl.h)."""

    _fields_ = [
        ("n_nodes", C.c_int32), ("episode_length", C.c_int32), ("arrival_mode", C.c_int32),
        ("disable_env_exploration", C.c_int32),
        ("node_cpu_capacity", C.c_double * EM_MAX_NODES),
        ("arrival_rate_lo", C.c_double), ("arrival_rate_hi", C.c_double), ("arrival_cycle_seconds", C.c_double),
        ("workload_lo", C.c_double), ("workload_hi", C.c_double),
        ("data_lo", C.c_double), ("data_hi", C.c_double),
        ("deadline_lo", C.c_double), ("deadline_hi", C.c_double),
        ("bw_lo", C.c_double), ("bw_hi", C.c_double),
        ("net_latency_lo", C.c_double), ("net_latency_hi", C.c_double),
        ("w_latency", C.c_double), ("w_energy", C.c_double), ("w_violation", C.c_double),
        ("min_time_step", C.c_double), ("max_time_step", C.c_double), ("initial_observation_window", C.c_double),
        ("initial_epsilon", C.c_double), ("final_epsilon", C.c_double),
        ("decay_steps", C.c_int64),
    ]


_ARRIVAL_MODES = {"cyclic": 1, "random": 2}


def make_em_config(config: dict) -> EmConfig:
    """Same defaults as EdgeSimGym.__init__ (MRRL.PY:128-148)."""
    c = EmConfig()
    n = int(config["n_nodes"])
    if not 1 <= n <= EM_MAX_NODES:
        raise ValueError(f"n_nodes must be in 1..{EM_MAX_NODES}")
    c.n_nodes = n
    c.episode_length = int(config["episode_length"])
    rate = config.get("task_arrival_rate", 10)
    lo, hi = config.get("task_arrival_rate_range", (rate, rate))
    c.arrival_rate_lo, c.arrival_rate_hi = float(lo), float(hi)
    c.arrival_mode = _ARRIVAL_MODES.get(config.get("task_arrival_mode", "cyclic"), 0)
    c.arrival_cycle_seconds = float(config.get("task_arrival_cycle_seconds", 20.0))
    c.disable_env_exploration = int(bool(config.get("disable_env_exploration", False)))
    caps = np.asarray(config["node_cpu_capacity"], dtype=np.float64).reshape(-1)
    if caps.size < n:
        raise ValueError("node_cpu_capacity shorter than n_nodes")
    for i in range(n):
        c.node_cpu_capacity[i] = float(caps[i])
    c.workload_lo, c.workload_hi = map(float, config["task_workload_range"])
    c.data_lo, c.data_hi = map(float, config["task_data_range"])
    c.deadline_lo, c.deadline_hi = map(float, config["task_deadline_range"])
    c.bw_lo, c.bw_hi = map(float, config["network_bw_range"])
    c.net_latency_lo, c.net_latency_hi = map(float, config["network_latency_range"])
    w = config["reward_weights"]
    c.w_latency, c.w_energy, c.w_violation = float(w["alpha"]), float(w["beta"]), float(w["gamma"])
    c.min_time_step = float(config.get("min_time_step", 0.02))
    c.max_time_step = float(config.get("max_time_step", 0.2))
    c.initial_observation_window = float(config.get("initial_observation_window", 0.1))
    sch = config.get("exploration_schedule", {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 1_000_000})
    c.initial_epsilon = float(sch.get("initial_epsilon", 1.0))
    c.final_epsilon = float(sch.get("final_epsilon", 0.05))
    c.decay_steps = int(sch.get("decay_steps", 1_000_000))
    return c


def mean_arrivals_bound(config: dict, margin: float = 1.25) -> float:
    """Upper bound on the MEAN number of task arrivals per environment step, times ``margin``: sizes the compact
    rollout storage of large batches (acting agents per step <= arrivals of the step before, MRRL.PY:353-362).

    A step draws dt = clip(Exp(mean 1/rate), a, b) and then n ~ Poisson(rate * dt) (MRRL.PY:244-247,350-351), so
    E[n | rate] = rate * E[dt] = rate*a + exp(-rate*a) - exp(-rate*b); the rate moves inside
    ``task_arrival_rate_range`` (MRRL.PY:220-235), the bound is the maximum over that range.  Default config: 0.99 (the
    measured mean of acting agents is 0.87, duplicates collapse); 10x the rate (50..150): 3.05."""
    rate = float(config.get("task_arrival_rate", 10))
    lo, hi = map(float, config.get("task_arrival_rate_range", (rate, rate)))
    a, b = float(config.get("min_time_step", 0.02)), float(config.get("max_time_step", 0.2))
    rates = np.linspace(max(min(lo, hi), 1e-6), max(lo, hi, 1e-6), 257)
    mean = rates * a + np.exp(-rates * a) - np.exp(-rates * b)
    first = rates * float(config.get("initial_observation_window", 0.1))     # reset's fixed window (MRRL.PY:209)
    return float(margin * max(mean.max(), 0.05)) + float(first.max()) / max(1, int(config.get("episode_length", 1000)))


def latency_log_capacity(config: dict, n_envs: int, per_step_cap: int) -> int:
    """Per-environment capacity of the latency log behind the exact p99 metric (MRRL.PY:476-479 keeps EVERY latency of
    the episode).  Small batches get the hard worst case (``per_step_cap`` tasks every step); large ones 1.5x the bound on
    the mean number of tasks of an episode, rounded up to a power of two (a convention kept from the LDS sort of round 1; the
    selection kernel that replaced it takes any length): 2,048 doubles =
    16 KB per environment at the default config, where an episode dispatches ~870 tasks with a standard deviation of ~30.
    An environment that overflows its log reports NaN for p99 (and only for p99); callers count such environments."""
    T = int(config.get("episode_length", 1000))
    full = T * int(per_step_cap)
    if int(n_envs) <= 1024:
        return full
    need = int(1.5 * mean_arrivals_bound(config, margin=1.0) * T) + 64
    cap = 1
    while cap < need:
        cap <<= 1
    return min(full, cap)


def summarise_episode_metrics(metrics, has_final):
    """Mean of the per-environment episode metrics as MAPPOTrainer.train_episode reports them (mappo_trainer.py:222-250):
    an environment whose final step produced no observation contributes 0.0 to every key (the reference only harvests
    ``infos`` of agents present in the last observation).  A metric that is MISSING (NaN: a p99 whose latency log
    overflowed or was switched off) is never turned into a number: it is left out of the mean and counted.
    Returns (means [6] float64 on the host, number of environments without an exact p99)."""
    import torch
    zero = torch.zeros((), dtype=metrics.dtype, device=metrics.device)
    per_env = torch.where(has_final.reshape(-1, 1).bool(), metrics, zero)
    missing = int(torch.isnan(per_env[:, 1]).sum().item())
    return torch.nanmean(per_env, dim=0).cpu().numpy(), missing


# ---------------------------------------------------------------------------------------------------------------
# EdgeMARLEnv (PyMARL twin environment): env_args of pymarl/src/config/envs/edge_marl.yaml:3-19
# ---------------------------------------------------------------------------------------------------------------
MARL_ENV_ARGS = {
    "n_nodes": 10,
    "node_cpu_capacity": [1.0e9, 1.2e9, 1.4e9, 1.6e9, 1.8e9, 2.0e9, 2.2e9, 2.4e9, 2.6e9, 3.0e9],
    "task_arrival_rate": 10.0,
    "task_arrival_rate_range": [5.0, 15.0],
    "task_arrival_mode": "cyclic",
    "task_arrival_cycle_seconds": 20.0,
    "task_workload_range": [1.0, 10.0],
    "task_data_range": [0.5, 5.0],
    "task_deadline_range": [0.05, 0.5],
    "network_bw_range": [10.0, 100.0],
    "network_latency_range": [0.002, 0.01],
    "reward_weights": {"alpha": 0.5, "beta": 0.3, "gamma": 0.2},
    "min_time_step": 0.02,
    "max_time_step": 0.2,
    "initial_observation_window": 0.1,
    "action_mode": "discrete",
    "episode_length": 1000,
}


class EmmConfig(C.Structure):
    """ctypes image of ``emm_config`` (include/edgemarl.h)."""

    _fields_ = [
        ("n_nodes", C.c_int32), ("episode_length", C.c_int32), ("arrival_mode", C.c_int32), ("action_mode", C.c_int32),
        ("node_cpu_capacity", C.c_double * EM_MAX_NODES),
        ("arrival_rate", C.c_double), ("arrival_rate_lo", C.c_double), ("arrival_rate_hi", C.c_double),
        ("arrival_cycle_seconds", C.c_double),
        ("workload_lo", C.c_double), ("workload_hi", C.c_double),
        ("data_lo", C.c_double), ("data_hi", C.c_double),
        ("deadline_lo", C.c_double), ("deadline_hi", C.c_double),
        ("bw_lo", C.c_double), ("bw_hi", C.c_double),
        ("net_latency_lo", C.c_double), ("net_latency_hi", C.c_double),
        ("alpha", C.c_double), ("beta", C.c_double), ("gamma", C.c_double),
        ("min_time_step", C.c_double), ("max_time_step", C.c_double),
    ]


def make_emm_config(env_args: dict) -> EmmConfig:
    """Same keys and defaults as EdgeMARLEnv.__init__ (edge_marl_env.py:18-41)."""
    c = EmmConfig()
    n = int(env_args.get("n_nodes", 10))
    if not 1 <= n <= EM_MAX_NODES:
        raise ValueError(f"n_nodes must be in 1..{EM_MAX_NODES}")
    c.n_nodes = n
    c.episode_length = int(env_args.get("episode_length", 1000))
    mode = env_args.get("action_mode", "discrete")
    if mode not in {"discrete", "hybrid"}:
        raise ValueError("EdgeMARLEnv action_mode must be 'discrete' or 'hybrid'.")      # edge_marl_env.py:40-41
    c.action_mode = 0 if mode == "discrete" else 1
    c.arrival_mode = _ARRIVAL_MODES.get(env_args.get("task_arrival_mode", "cyclic"), 0)
    c.arrival_rate = float(env_args.get("task_arrival_rate", 10.0))
    c.arrival_rate_lo, c.arrival_rate_hi = map(float, env_args.get("task_arrival_rate_range", (5.0, 15.0)))
    c.arrival_cycle_seconds = float(env_args.get("task_arrival_cycle_seconds", 20.0))
    caps = np.asarray(env_args.get("node_cpu_capacity", np.ones(n) * 1e9), dtype=np.float64).reshape(-1)
    if caps.size < n:
        raise ValueError("node_cpu_capacity shorter than n_nodes")
    for i in range(n):
        c.node_cpu_capacity[i] = float(caps[i])
    c.workload_lo, c.workload_hi = map(float, env_args.get("task_workload_range", (1.0, 10.0)))
    c.data_lo, c.data_hi = map(float, env_args.get("task_data_range", (0.5, 5.0)))
    c.deadline_lo, c.deadline_hi = map(float, env_args.get("task_deadline_range", (0.05, 0.5)))
    c.bw_lo, c.bw_hi = map(float, env_args.get("network_bw_range", (10.0, 100.0)))
    c.net_latency_lo, c.net_latency_hi = map(float, env_args.get("network_latency_range", (0.002, 0.01)))
    w = env_args.get("reward_weights", {"alpha": 0.5, "beta": 0.3, "gamma": 0.2})
    c.alpha, c.beta, c.gamma = float(w.get("alpha", 0.5)), float(w.get("beta", 0.3)), float(w.get("gamma", 0.2))
    c.min_time_step = float(env_args.get("min_time_step", 0.02))
    c.max_time_step = float(env_args.get("max_time_step", 0.2))
    return c
