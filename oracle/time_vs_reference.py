"""Container-only tool: times the IMPORTED reference and the repository's CPU restatements side by side on the same
workload, so that `cpu_baseline.kind = "port"` in bench.py is shown to be representative.
TEST INFRASTRUCTURE; needs /root/reference, never runs on the GPU box.

Three legs, one process, same seeds, same host:
  env      1000-step episodes, random dict actions, exploration off:  EdgeSimGym.step  vs  EdgeEnvOracle.step
  rollout  MAPPOTrainer.train_episode(explore=True)                  vs  mappo_loop_oracle.run_episode
  update   MAPPO.update() on that episode's buffer                    vs  mappo_loop_oracle.update_from_buffer
Writes profiles/port_vs_reference.json = {"env": {"reference": r, "port": p, "ratio": p / r}, ...} (rates in the unit
named there; ratio > 1 means the port is FASTER than the reference, i.e. the CPU baseline printed by bench.py flatters the
CPU by that factor).

    python oracle/time_vs_reference.py [seconds per leg, default 8]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import edge_env_oracle as eo  # noqa: E402
from oracle import mappo_loop_oracle as loop  # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402


def time_env_reference(M, seconds):
    cfg = dict(M.ENV_CONFIG)
    cfg["disable_env_exploration"] = True
    env = M.EdgeSimGym(cfg)
    n = env.n_nodes
    rs = np.random.RandomState(1)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        obs, _ = env.reset(seed=steps)
        for _ in range(cfg["episode_length"]):
            pl, rr = rs.randint(0, n, size=n), rs.uniform(0.1, 1.0, size=n).astype(np.float32)
            act = {a: {"placement": int(pl[int(a.split("_")[1])]), "resource": rr[int(a.split("_")[1]):int(a.split("_")[1]) + 1]}
                   for a in obs}
            obs, _, _, _, _ = env.step(act)
        steps += cfg["episode_length"]
    return steps * n / (time.perf_counter() - t0)


def time_env_port(seconds):
    cfg = eo.default_config()
    cfg["disable_env_exploration"] = True
    env = eo.EdgeEnvOracle(cfg, eo.LegacyDraws(0))
    n = env.n
    rs = np.random.RandomState(1)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.reset()
        for _ in range(cfg["episode_length"]):
            env.step(rs.randint(0, n, size=n), rs.uniform(0.1, 1.0, size=n).astype(np.float32))
        steps += cfg["episode_length"]
    return steps * n / (time.perf_counter() - t0)


def time_loop_reference(M, T, seconds):
    tr = T.MAPPOTrainer(M.ENV_CONFIG, seed=0, device="cpu")
    t_roll = t_upd = 0.0
    trans = steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        a = time.perf_counter()
        st = tr.train_episode(explore=True, record_buffer=True)
        b = time.perf_counter()
        L = len(tr.mappo.buffer.rewards)
        tr.mappo.update()
        c = time.perf_counter()
        t_roll += b - a
        t_upd += c - b
        trans += L
        steps += st["episode_steps"]
    return {"rollout_env_steps_per_s": steps / t_roll, "update_transitions_per_s": trans / t_upd,
            "loop_transitions_per_s": trans / (t_roll + t_upd)}


def time_loop_port(seconds):
    cfg = eo.default_config()
    env = eo.EdgeEnvOracle(cfg, eo.LegacyDraws(0))
    params = loop.init_params(env.n, 0)
    t_roll = t_upd = 0.0
    trans = steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        a = time.perf_counter()
        buf = loop.run_episode(env, params, explore=True)
        b = time.perf_counter()
        if buf["obs"]:
            params, _ = loop.update_from_buffer(params, buf)
        c = time.perf_counter()
        t_roll += b - a
        t_upd += c - b
        trans += len(buf["obs"])
        steps += cfg["episode_length"]
    return {"rollout_env_steps_per_s": steps / t_roll, "update_transitions_per_s": trans / t_upd,
            "loop_transitions_per_s": trans / (t_roll + t_upd)}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
    M, _, T = load_reference()
    out = {"host": {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads()}, "seconds_per_leg": seconds}
    r, p = time_env_reference(M, seconds), time_env_port(seconds)
    out["env"] = {"unit": "agent-steps/s", "reference": r, "port": p, "ratio": p / r}
    lr, lp = time_loop_reference(M, T, 2 * seconds), time_loop_port(2 * seconds)
    for key, unit in (("rollout_env_steps_per_s", "env-steps/s"), ("update_transitions_per_s", "transitions/s"),
                      ("loop_transitions_per_s", "transitions/s")):
        out[key.rsplit("_per_s", 1)[0]] = {"unit": unit, "reference": lr[key], "port": lp[key], "ratio": lp[key] / lr[key]}
    path = os.path.join(ROOT, "profiles", "port_vs_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
