"""Greedy baseline on the batched environment (reference ``run_greedy_baseline``, MRRL.PY:778-846): every acting
agent places its task on the node with the smallest load/capacity ratio and asks for half of its CPU."""
from __future__ import annotations

import numpy as np
import torch

from .config import ENV_CONFIG
from .env import METRIC_KEYS, BatchedEdgeSimGym


def run_greedy_baseline(seed, num_eval_episodes=100, n_envs=None, device="cuda:0", config=None):
    """Returns one log row with the reference's keys; episodes are played ``n_envs`` at a time."""
    cfg = {**(ENV_CONFIG if config is None else config), "disable_env_exploration": True}
    B = int(n_envs or min(int(num_eval_episodes), 4096))
    env = BatchedEdgeSimGym(cfg, n_envs=B, device=device, seed=seed)
    n = env.n_nodes
    caps = torch.as_tensor(env.cpu_caps, dtype=torch.float64, device=env.device)
    resource = torch.full((B, n), 0.5, dtype=torch.float32, device=env.device)
    rows, rewards = [], []
    played = 0
    while played < num_eval_episodes:
        obs, _ = env.reset(seed=seed + played) if played == 0 else env.reset()
        total = torch.zeros(B, dtype=torch.float64, device=env.device)
        for _ in range(env.episode_len):
            # argmin(nodes_load / cpu_caps), MRRL.PY:808: the load ratios are what each observation row carries
            # (own ratio in column 0, the others in columns 5..), so rebuild them from the state copy
            load = torch.as_tensor(env.get_state()["load"], device=env.device)
            target = torch.argmin(load / caps, dim=1).to(torch.int32)
            placement = target.unsqueeze(1).expand(B, n).contiguous()
            obs, rew, done, _, _ = env.step(placement, resource)
            total += rew
        m = env.episode_metrics().cpu().numpy()
        take = min(B, num_eval_episodes - played)
        rows.append(m[:take])
        rewards.append(total[:take].cpu().numpy())
        played += take
    m = np.concatenate(rows)
    out = {"algorithm": "Greedy", "seed": seed, "iteration": 0, "timestep": 0,
           "episode_return_mean": float(np.concatenate(rewards).mean())}
    for i, k in enumerate(METRIC_KEYS):
        if k != "violation_rate":
            out[k] = float(np.nanmean(m[:, i]))
    return [out]
