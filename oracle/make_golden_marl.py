"""Container-only tool: record golden episodes of the reference's PyMARL twin environment (EdgeMARLEnv).

TEST INFRASTRUCTURE.  Run as ``python -m oracle.make_golden_marl`` inside the build container (needs /root/reference).
Imports ``pymarl/src/envs/edge_marl_env.py`` under a throw-away package name (its directory's ``__init__`` pulls in
PyMARL's registry and is not executed), swaps the environment's private ``numpy.random.Generator`` for a recording
proxy, drives scripted actions and stores, per fixture: the configuration, the actions, the flat draw log, and the
reference's per-step outputs (reward, terminated, the five info scalars, the observation tensor) plus its episode
summary.  ``tests/test_oracle_marl_env.py`` replays the log through ``oracle/edge_marl_oracle.py`` (bit-exact pin) and
converts it into the structured tape the HIP kernel consumes.  Only data is written: no reference source text.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import types

import numpy as np

REFERENCE_ENVS = "/root/reference/pymarl/src/envs"
OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
INFO_KEYS = ("avg_latency_ms", "avg_energy_J", "deadline_violation_rate", "throughput_tps", "load_balance_jain")
SUMMARY_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "deadline_violation_rate", "load_balance_jain")


def load_reference_env():
    if not os.path.isdir(REFERENCE_ENVS):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("_ref_pymarl_envs")
    pkg.__path__ = [REFERENCE_ENVS]
    sys.modules["_ref_pymarl_envs"] = pkg
    return importlib.import_module("_ref_pymarl_envs.edge_marl_env").EdgeMARLEnv


class RecordingGenerator:
    """Wraps the environment's numpy Generator; logs (name, args, result) of every scalar draw."""

    def __init__(self, gen):
        self._gen, self.log = gen, []

    def _wrap(self, name, *args):
        out = getattr(self._gen, name)(*args)
        self.log.append((name, tuple(float(a) for a in args), float(out)))
        return out

    def uniform(self, *args):
        return self._wrap("uniform", *args)

    def exponential(self, *args):
        return self._wrap("exponential", *args)


def scripted_actions(kind, n, steps, seed):
    rng = np.random.RandomState(seed)
    if kind == "discrete":
        a = rng.randint(0, n, size=(steps, n))
        a[::7] = np.arange(n)                       # all-local steps (no network draws)
        a[3::11] = 0                                # everyone piles onto server 0 (ordered load chain)
        a[5::13, 0] = -3                            # out-of-range values are clipped (edge_marl_env.py:101)
        a[6::13, 1] = n + 5
        return a.astype(np.int32)
    a = rng.uniform(-1.5, 1.5, size=(steps, n, n + 1)).astype(np.float32)
    a[::9, :, :n] = 0.0                             # all-zero logits -> placement 0 (edge_marl_env.py:311-312)
    a[4::10, 2, :n] = 5e-9                          # inside allclose's absolute tolerance
    a[2::6, :, -1] = 3.0                            # resource scalar beyond [-1, 1] is clipped
    return a


def record(name, args, kind, steps, action_seed):
    Env = load_reference_env()
    np.random.seed(12345)                           # the env must not touch the global stream; kept fixed anyway
    env = Env.__new__(Env)
    # __init__ resets once with the generator it builds from the seed; the proxy is installed right after and the
    # episode that is recorded starts with an explicit reset(), exactly as PyMARL's runner does
    Env.__init__(env, **args)
    rec = RecordingGenerator(env.rng)
    env.rng = rec
    env.reset()
    n = env.n_agents
    actions = scripted_actions(kind, n, steps, action_seed)
    obs = [np.stack(env.get_obs())]
    state0 = env.get_state()
    rewards, terms, infos = [], [], []
    for t in range(steps):
        acts = list(actions[t]) if kind == "discrete" else [actions[t, i] for i in range(n)]
        r, term, info = env.step(acts)
        rewards.append(r); terms.append(term); infos.append([info[k] for k in INFO_KEYS])
        obs.append(np.stack(env.get_obs()))
        assert info["episode_limit"] == term
    summary = env.get_episode_summary()
    log = rec.log
    names = sorted({e[0] for e in log})
    out = {
        "config_json": np.frombuffer(json.dumps(args, sort_keys=True).encode(), dtype=np.uint8),
        "actions": actions,
        "draw_kind": np.array([names.index(e[0]) for e in log], dtype=np.int8),
        "draw_names": np.array(names),
        "draw_args": np.array([list(e[1]) + [np.nan] * (2 - len(e[1])) for e in log], dtype=np.float64),
        "draw_value": np.array([e[2] for e in log], dtype=np.float64),
        "obs": np.stack(obs).astype(np.float32),
        "state0": state0.astype(np.float32),
        "reward": np.array(rewards, dtype=np.float64),
        "terminated": np.array(terms, dtype=np.uint8),
        "info": np.array(infos, dtype=np.float64),
        "summary": np.array([summary[k] for k in SUMMARY_KEYS], dtype=np.float64),
        "sizes": np.array([env.get_obs_size(), env.get_state_size(), env.get_total_actions(), env.get_action_dim()], dtype=np.int64),
    }
    path = os.path.join(OUT_DIR, f"marlenv_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{path}: {steps} steps, {len(log)} draws, {os.path.getsize(path)} bytes")


def main():
    from .edge_marl_oracle import default_args

    base = default_args()
    base["seed"] = 7
    base["episode_length"] = 240
    record("cyclic_discrete", dict(base), "discrete", 240, 1)
    rnd = dict(base, task_arrival_mode="random", action_mode="hybrid", seed=11, episode_length=120)
    record("random_hybrid", rnd, "hybrid", 120, 2)
    small = dict(base, n_nodes=4, node_cpu_capacity=[1.0e9, 2.0e9, 1.5e9, 3.0e9], task_arrival_mode="fixed",
                 task_arrival_rate=25.0, seed=3, episode_length=60,
                 reward_weights={"alpha": 0.4, "beta": 0.4, "gamma": 0.2})
    record("fixed_small", small, "discrete", 60, 3)
    wide = dict(base, n_nodes=24, node_cpu_capacity=list(np.linspace(1.0e9, 3.0e9, 24)), seed=5, episode_length=40)
    record("cyclic_wide", wide, "discrete", 40, 4)


if __name__ == "__main__":
    main()
