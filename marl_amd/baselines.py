"""Greedy baseline on the batched environment (reference ``run_greedy_baseline``, MRRL.PY:778-846): every acting
agent places its task on the node with the smallest load/capacity ratio and asks for half of its CPU.  The policy
runs inside the step kernel (``em_step_greedy``: shuffle argmin over the lanes of an environment), so an episode
batch needs no host round trip."""
from __future__ import annotations

import numpy as np
import torch

from .config import ENV_CONFIG
from .env import METRIC_KEYS, BatchedEdgeSimGym

_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain")


def run_greedy_baseline(seed, num_episodes=100, n_envs=None, device="cuda:0", config=None):
    """Returns the reference's result dict (``<metric>_mean`` / ``<metric>_std`` over the evaluated episodes,
    MRRL.PY:829-842).  Episodes are played ``n_envs`` at a time; like the reference, an episode contributes only
    when its final step produced observations (``if done and obs``, MRRL.PY:817)."""
    cfg = {**(ENV_CONFIG if config is None else config), "disable_env_exploration": True}
    B = int(n_envs or min(int(num_episodes), 4096))
    env = BatchedEdgeSimGym(cfg, n_envs=B, device=device, seed=seed, latency_log_capacity=int(cfg["episode_length"]) * int(cfg["n_nodes"]))
    rows = []
    played = 0
    first = True
    while played < num_episodes:
        env.reset(seed=seed + played) if first else env.reset()
        first = False
        env.step_many(env.episode_len, greedy_share=0.5)     # the whole episode in one launch, policy in the kernel
        m = env.episode_metrics()
        has_final = env.active != 0
        take = min(B, num_episodes - played)
        rows.append(m[:take][has_final[:take]].cpu().numpy())
        played += take
    env.close()
    m = np.concatenate(rows) if rows else np.zeros((0, 6))
    idx = {k: i for i, k in enumerate(METRIC_KEYS)}
    out = {"algorithm": "Greedy", "seed": seed}
    for k in _KEYS:
        col = m[:, idx[k]]
        out[f"{k}_mean"] = float(np.mean(col)) if col.size else float("nan")
        out[f"{k}_std"] = float(np.std(col)) if col.size else float("nan")
    out["episodes_counted"] = int(m.shape[0])
    return out
