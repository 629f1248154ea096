"""Regenerates tests/golden/*.npz from the IMPORTED reference.  Build-container only.

    python -m oracle.make_golden            # writes every fixture

For each scenario the reference ``EdgeSimGym`` (``MRRL.PY:114-521``) is run with its module-level
``np`` replaced by a recording proxy, so every legacy-RNG draw is logged.  The restatement in
``oracle/edge_env_oracle.py`` is then replayed on that log (which also checks that it asks for the
same draws with bit-identical arguments in the same order), must reproduce the reference's outputs
bit for bit, and emits the structured tape that the HIP kernels consume.  Fixtures hold data only:
tapes (inputs) and the reference's outputs.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import edge_env_oracle as eo  # noqa: E402
from oracle.ref_loader import NumpyProxy, load_reference  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")

METRIC_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "violation_rate", "throughput_tps",
               "load_balance_jain")


def policy_actions(rs, n, mode):
    """Pre-drawn policy actions from a PRIVATE RandomState (never the global stream)."""
    if mode == "fixed":
        return np.full(n, 3, np.int32), np.full(n, 0.5, np.float32)
    return rs.randint(0, n, size=n).astype(np.int32), rs.uniform(0.1, 1.0, size=n).astype(np.float32)


def run_scenario(name, seed, cfg_over, action_mode, explore_counter0=0):
    MRRL, _, _ = load_reference()
    real_np = np
    proxy = NumpyProxy(real_np)
    MRRL.np = proxy
    try:
        cfg = {**MRRL.ENV_CONFIG, **cfg_over}
        n = cfg["n_nodes"]
        T = cfg["episode_length"]
        MRRL.EdgeSimGym._global_step_counter = explore_counter0
        env = MRRL.EdgeSimGym(cfg)
        act_rs = real_np.random.RandomState(seed + 7919)
        placements = np.zeros((T, n), np.int32)
        resources = np.zeros((T, n), np.float32)

        obs_all = np.zeros((T + 1, n, n + 4), np.float32)
        rank_all = np.full((T + 1, n), -1, np.int8)
        reward = np.zeros(T)
        emitted = np.zeros(T, bool)
        done = np.zeros(T, bool)
        load = np.zeros((T + 1, n))
        queue = np.zeros((T + 1, n))
        sim_time = np.zeros(T + 1)
        step_lat = np.full((T, n), np.nan)
        n_viol = np.zeros(T + 1, np.int64)
        n_tasks = np.zeros(T + 1, np.int64)
        n_completed = np.zeros(T + 1, np.int64)

        def put(i, od):
            for r, (k, v) in enumerate(od.items()):
                a = int(k.split("_")[1])
                obs_all[i, a] = v
                rank_all[i, a] = r

        od, _ = env.reset(seed=seed)
        put(0, od)
        load[0], queue[0], sim_time[0] = env.nodes_load, env.nodes_queue_len, env.sim_time
        metrics = None
        for t in range(T):
            pl, rsrc = policy_actions(act_rs, n, action_mode)
            placements[t], resources[t] = pl, rsrc
            actions = {k: {"placement": int(pl[int(k.split("_")[1])]),
                           "resource": np.array([rsrc[int(k.split("_")[1])]], dtype=np.float32)} for k in od}
            n_before = len(env.episode_latencies)
            viol_before = env.episode_violations
            od, rew, dn, _, infos = env.step(actions)
            put(t + 1, od)
            emitted[t] = bool(rew)
            if rew:
                reward[t] = list(rew.values())[0]
            else:
                # the reference computed the reward but did not emit it (MRRL.PY:452-455); recompute it with
                # the reference's own function (np.sum only sees the number of violations)
                k_new = len(env.episode_latencies) - n_before
                dv = env.episode_violations - viol_before
                reward[t] = env._calculate_reward(env.episode_latencies[n_before:], env.episode_energies[n_before:],
                                                  [1] * dv + [0] * (k_new - dv))
            done[t] = dn["__all__"]
            load[t + 1], queue[t + 1], sim_time[t + 1] = env.nodes_load, env.nodes_queue_len, env.sim_time
            new_lat = env.episode_latencies[n_before:]
            step_lat[t, :len(new_lat)] = new_lat
            n_viol[t + 1] = env.episode_violations
            n_tasks[t + 1] = len(env.episode_latencies)
            n_completed[t + 1] = env.episode_completed_tasks
            if dn["__all__"]:
                holder = infos.get("__common__") or next(v for v in infos.values() if "episode_metrics" in v)
                metrics = holder["episode_metrics"]
        flat = proxy.random.log
        explore_counter1 = MRRL.EdgeSimGym._global_step_counter
    finally:
        MRRL.np = real_np

    # replay on the restatement: checks draw order/arguments and emits the structured tape
    tape = eo.empty_tape(T + 1, n, a_slots=32)
    ocfg = {**eo.default_config(), **cfg_over}
    orc = eo.EdgeEnvOracle(ocfg, eo.FlatReplay(flat), tape_out=tape)
    orc.explore_counter = explore_counter0
    o_obs, o_order = orc.reset(seed)
    assert np.array_equal(o_obs, obs_all[0]), name
    for t in range(T):
        out = orc.step(placements[t], resources[t])
        assert np.array_equal(out.obs, obs_all[t + 1]), (name, t)
        assert [int(a) for a in out.order] == [int(np.where(rank_all[t + 1] == r)[0][0]) for r in range((rank_all[t + 1] >= 0).sum())], (name, t)
        assert out.reward_emitted == emitted[t], (name, t)
        assert out.reward == reward[t], (name, t, out.reward, reward[t])
        assert np.array_equal(orc.load, load[t + 1]) and np.array_equal(orc.queue, queue[t + 1]), (name, t)
        assert orc.sim_time == sim_time[t + 1]
    assert orc.draws.pos == len(flat), "restatement consumed fewer draws than the reference"
    assert orc.explore_counter == explore_counter1
    om = orc.episode_metrics()
    for k in METRIC_KEYS:
        assert om[k] == metrics[k], (name, k, om[k], metrics[k])

    out_path = os.path.join(GOLDEN, f"env_{name}.npz")
    np.savez_compressed(
        out_path,
        seed=seed, n_nodes=n, episode_length=T, explore_counter0=explore_counter0,
        disable_env_exploration=bool(cfg.get("disable_env_exploration", False)),
        decay_steps=cfg["exploration_schedule"]["decay_steps"],
        arrival_mode=cfg["task_arrival_mode"],
        rate_range=np.array(cfg["task_arrival_rate_range"], np.float64),
        placements=placements, resources=resources,
        obs=obs_all, rank=rank_all, reward=reward, reward_emitted=emitted, done=done,
        load=load, queue=queue, sim_time=sim_time, step_lat=step_lat,
        n_viol=n_viol, n_tasks=n_tasks, n_completed=n_completed,
        metrics=np.array([metrics[k] for k in METRIC_KEYS]),
        **{f"tape_{k}": v for k, v in tape.items()},
    )
    print(f"[golden] {out_path}: {len(flat)} draws, sum(reward*emitted)={float((reward * emitted).sum())!r}")


SCENARIOS = [
    # name, seed, config overrides, action mode, initial exploration counter
    ("seed0_fixed_noexplore", 0, {"disable_env_exploration": True}, "fixed", 0),
    ("seed1_rand_noexplore_T300", 1, {"disable_env_exploration": True, "episode_length": 300}, "random", 0),
    ("seed2_rand_explore_T300", 2, {"episode_length": 300,
                                    "exploration_schedule": {"initial_epsilon": 1.0, "final_epsilon": 0.05,
                                                             "decay_steps": 150}}, "random", 0),
    ("seed3_rand_explore_late_T200", 3, {"episode_length": 200}, "random", 600_000),
    ("seed4_randomrate_T200", 4, {"episode_length": 200, "task_arrival_mode": "random",
                                  "disable_env_exploration": True}, "random", 0),
    ("seed5_heavy_T120", 5, {"episode_length": 120, "task_arrival_rate_range": (30, 45),
                             "disable_env_exploration": True}, "random", 0),
]


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    for sc in SCENARIOS:
        run_scenario(*sc)


if __name__ == "__main__":
    main()
