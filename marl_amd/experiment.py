"""``MRRL.run_experiment`` (MRRL.PY:573-775): the reference's per-(algorithm, seed) training entry that
``quick_test_5_algorithms.py:32-63`` and ``MRRL.main`` call.  The reference dispatches "MAPPO" to its hand-written trainer
(:578-580) and builds "IPPO" from RLlib's PPO with one policy per agent (:611-638; ``ray[rllib]`` is not vendored, parity
unpinned); here both branches run on the HIP kernels: MAPPO through :func:`marl_amd.trainer.run_mappo_experiment`, IPPO
through :class:`marl_amd.ippo.IPPOTrainer` on the reference topology (deciders == servers).  "MADDPG"/"QMIX" raise
``NotImplementedError`` exactly as the reference does (:597-600: they live in PyMARL).  "Greedy" is an extension of the
dispatcher (the reference calls ``run_greedy_baseline`` directly, quick_test_5_algorithms.py:60)."""
from __future__ import annotations

import math

from .baselines import run_greedy_baseline
from .config import ENV_CONFIG
from .trainer import run_mappo_experiment

METRIC_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain")   # MRRL.PY:602


def run_experiment(alg_name, seed, num_iterations=None, max_timesteps=5_000_000, eval_interval_steps=50_000, n_envs=1,
                   device="cuda:0", env_config=None, eval_episodes=10, verbose=True):
    """Returns the list of log entries ``{"algorithm", "seed", "timestep", "episode_return_mean"[, the five evaluation
    metrics]}`` (MRRL.PY:720-745).  ``timestep`` counts steps of ONE environment, as ``MAPPOTrainer.train`` does; the batch
    plays ``n_envs`` times as many.  Like the reference, the step budget rules and ``num_iterations`` is only a safety cap
    that is raised when it cannot cover ``max_timesteps`` (:707-716)."""
    if alg_name == "MAPPO":                                             # MRRL.PY:578-580
        return run_mappo_experiment(seed, num_iterations if num_iterations is not None else 1000,
                                    max_total_steps=max_timesteps, n_envs=n_envs, device=device)
    if alg_name in ("MADDPG", "QMIX"):                                  # MRRL.PY:597-600
        raise NotImplementedError(f"{alg_name} runs in the PyMARL stack (pymarl_configs/README.md); it is not part of this path")
    if alg_name == "Greedy":
        return [run_greedy_baseline(seed, num_episodes=eval_episodes if num_iterations is None else int(num_iterations),
                                    device=device, config=env_config)]
    if alg_name != "IPPO":
        raise ValueError(f"unknown algorithm: {alg_name}")              # MRRL.PY:659-660

    from .ippo import IPPOTrainer
    cfg = {**(ENV_CONFIG if env_config is None else env_config), "disable_env_exploration": False}      # MRRL.PY:604-607
    trainer = IPPOTrainer(cfg, seed=seed, device=device, n_envs=n_envs, verbose=verbose)
    T = trainer.env.episode_len
    expected = math.ceil(max_timesteps / max(T, 1))                     # one iteration = one episode per environment
    cap = None if num_iterations is None else max(int(num_iterations), expected)
    eval_every = max(1, math.ceil(eval_interval_steps / T))             # MRRL.PY:676
    log, steps, it = [], 0, 0
    while True:
        st = trainer.train_episode(explore=True, record_buffer=True)
        trainer.algo.update()
        it += 1
        steps += st["episode_steps"]
        entry = {"algorithm": alg_name, "seed": seed, "timestep": steps, "episode_return_mean": st["episode_reward"]}
        if it % eval_every == 0:
            ev = trainer.evaluate(num_episodes=eval_episodes)           # explore off, env exploration off (:686-689)
            entry.update({k: ev[k] for k in METRIC_KEYS})
        log.append(entry)
        if verbose:
            lat = entry.get("avg_latency_ms")
            print(f"Alg: {alg_name}, Seed: {seed}, Iter: {it}, Steps: {steps}, Train Reward: {entry['episode_return_mean']:.2f}, "
                  f"Eval Latency(ms): {'N/A' if lat is None else format(lat, '.2f')}")
        if steps >= max_timesteps or (cap is not None and it >= cap):
            break
    trainer.env.close()
    return log
