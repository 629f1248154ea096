"""CPU port of the reference's env-plus-MAPPO loop for timing.  TEST INFRASTRUCTURE ONLY.

Same structure as the reference: ``MAPPOTrainer.train_episode`` (``mappo_trainer.py:177-250``) with one tiny
torch-CPU forward per acting agent (``mappo_algorithm.py:189-224``), one critic forward per step (:226-239), a
python-list buffer with one entry per acting agent (``mappo_trainer.py:160-175``), then ``MAPPO.update``
(:264-345) = sequential float64 GAE + 4 full-batch epochs.  Built from the pinned restatements
(``edge_env_oracle.EdgeEnvOracle``, ``mappo_oracle``).  Only ``bench.py``'s ``cpu_baseline`` leg and tests use it.
"""
from __future__ import annotations

import time

import numpy as np
import torch

from . import edge_env_oracle as eo
from . import mappo_oracle as mo


def init_params(n: int, seed: int = 0) -> dict:
    """nn.Linear default init in the reference's construction order (actor, then critic)."""
    torch.manual_seed(seed)
    obs_dim, act_dim, state_dim = n + 4, n + 1, n * (n + 4)
    a1, a2, a3 = torch.nn.Linear(obs_dim, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, act_dim)
    c1, c2, c3 = torch.nn.Linear(state_dim, 64), torch.nn.Linear(64, 64), torch.nn.Linear(64, 1)
    p = {"actor.fc1.weight": a1.weight, "actor.fc1.bias": a1.bias, "actor.fc2.weight": a2.weight,
         "actor.fc2.bias": a2.bias, "actor.mu.weight": a3.weight, "actor.mu.bias": a3.bias,
         "actor.log_std": torch.zeros(act_dim),
         "critic.fc1.weight": c1.weight, "critic.fc1.bias": c1.bias, "critic.fc2.weight": c2.weight,
         "critic.fc2.bias": c2.bias, "critic.value_out.weight": c3.weight, "critic.value_out.bias": c3.bias}
    return {k: v.detach().numpy().copy() for k, v in p.items()}


def run_episode(env: eo.EdgeEnvOracle, params: dict, explore: bool = True):
    """One episode; returns the flattened per-agent buffer and the phase time of the rollout."""
    n = env.n
    p = mo.to_torch(params)
    buf = {"obs": [], "states": [], "actions": [], "old_logp": [], "rewards": [], "values": [], "dones": []}
    obs, order = env.reset()
    done = False
    with torch.no_grad():
        while not done:
            state = torch.as_tensor(obs.reshape(1, -1))
            placement = np.zeros(n, dtype=np.int64)
            resource = np.full(n, 0.5, dtype=np.float32)
            acts, logps = {}, {}
            for agent in order:                                  # one forward per acting agent, as the reference
                o = torch.as_tensor(obs[agent]).reshape(1, -1)
                if explore:
                    a, lp, _ = mo.sample(p, o, torch.randn(1, n + 1))
                else:
                    a, lp = mo.actor_mu(p, o), torch.zeros(1)
                acts[agent], logps[agent] = a[0].numpy(), float(lp[0])
                placement[agent], resource[agent] = mo.format_env_action(acts[agent], n)
            value = float(mo.rollout_value(p, state)[0])
            out = env.step(placement, resource)
            reward = out.reward if out.reward_emitted else 0.0   # mappo_trainer.py:209-212
            for agent in order:
                buf["obs"].append(obs[agent].copy())
                buf["states"].append(state[0].numpy())
                buf["actions"].append(acts[agent])
                buf["old_logp"].append(logps[agent])
                buf["rewards"].append(reward)
                buf["values"].append(value)
                buf["dones"].append(out.done)
            obs, order, done = out.obs, out.order, out.done
    return buf


def update_from_buffer(params: dict, buf: dict, n_iter: int = 4):
    adv, ret = mo.gae_flat(buf["rewards"], buf["values"], buf["dones"])
    batch = {"obs": np.asarray(buf["obs"], np.float32), "states": np.asarray(buf["states"], np.float32),
             "actions": np.asarray(buf["actions"], np.float32), "old_logp": buf["old_logp"], "adv": adv, "returns": ret}
    new_params, losses, _, _, _ = mo.update(params, batch, n_iter=n_iter)
    return new_params, losses


def time_loop(seconds_target: float = 10.0, seed: int = 0) -> dict:
    """Whole episodes of rollout + update until ~seconds_target; returns transitions/s and env-steps/s."""
    cfg = eo.default_config()
    env = eo.EdgeEnvOracle(cfg, eo.LegacyDraws(seed))
    params = init_params(env.n, seed)
    transitions = steps = episodes = 0
    t_roll = t_upd = 0.0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        a = time.perf_counter()
        buf = run_episode(env, params, explore=True)
        b = time.perf_counter()
        if buf["obs"]:
            params, _ = update_from_buffer(params, buf)
        c = time.perf_counter()
        t_roll += b - a
        t_upd += c - b
        transitions += len(buf["obs"])
        steps += cfg["episode_length"]
        episodes += 1
    dt = time.perf_counter() - t0
    return {"samples_per_s": transitions / dt, "env_steps_per_s": steps / dt, "episodes": episodes, "seconds": dt,
            "rollout_s": t_roll, "update_s": t_upd, "torch_threads": torch.get_num_threads()}
