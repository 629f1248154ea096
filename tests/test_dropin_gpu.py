"""The drop-in surface: the reference's module names, the dict-API EdgeSimGym (n_envs = 1 view of the kernels) and
the calls its smoke scripts make (test_real_mappo.py:16-38, quick_test_5_algorithms.py, mappo_trainer.py:397-409)."""
import numpy as np
import pytest

from oracle import edge_env_oracle as eo

pytestmark = pytest.mark.gpu


def _oracle(cfg, seed):
    return eo.DictEnvOracle(cfg, eo.PhiloxDraws(seed, 0, episode=0))


@pytest.mark.parametrize("vector_actions", [False, True])
def test_dict_api_matches_dict_oracle(vector_actions):
    """MRRL.EdgeSimGym reset()/step(action_dict): same keys in the same (arrival) order, same observations, rewards,
    done flags and episode metrics as the dict adapter of the numpy restatement on the same Philox stream."""
    import MRRL
    cfg = dict(MRRL.ENV_CONFIG, episode_length=60, disable_env_exploration=True)
    seed = 123
    env = MRRL.EdgeSimGym(cfg, seed=seed)
    ocfg = dict(eo.default_config(), episode_length=60, disable_env_exploration=True)
    ref = _oracle(ocfg, seed)
    ref.core.draws.episode = 1          # the kernels number episodes from 1 (first reset)
    obs, info = env.reset()
    r_obs, _ = ref.reset()
    rs = np.random.RandomState(5)
    n = env.n_nodes
    for t in range(60):
        assert list(obs.keys()) == list(r_obs.keys()), t
        for k in obs:
            assert np.array_equal(obs[k], r_obs[k]), (t, k)
        if vector_actions:
            acts = {f"node_{a}": rs.normal(0, 1, n + 1).astype(np.float32) for a in range(n)}
            r_acts = {}
            for k in r_obs:
                pl, res = eo.convert_action(acts[k], n)
                r_acts[k] = {"placement": pl, "resource": np.array([res], np.float32)}
        else:
            acts = {k: {"placement": int(rs.randint(0, n)), "resource": np.array([rs.uniform(0.1, 1.0)], np.float32)}
                    for k in obs}
            r_acts = acts
        obs, rew, dones, trunc, infos = env.step(acts)
        r_obs, r_rew, r_dones, _, r_infos = ref.step(r_acts)
        assert rew == r_rew and dones == r_dones and trunc == {"__all__": False}, t
    assert dones["__all__"]
    holder = infos.get("__common__") or next(v for v in infos.values() if "episode_metrics" in v)
    r_holder = r_infos.get("__common__") or next(v for v in r_infos.values() if "episode_metrics" in v)
    for k, v in r_holder["episode_metrics"].items():
        np.testing.assert_allclose(holder["episode_metrics"][k], v, rtol=1e-12, atol=0, err_msg=k)
    assert env.nodes_load.shape == (n,) and env.cpu_caps.shape == (n,)
    # the exploration switch and the class-level step counter of the reference (MRRL.PY:122,179)
    env.disable_env_exploration = False
    c0 = MRRL.EdgeSimGym._global_step_counter
    env.reset()
    env.step({})
    assert MRRL.EdgeSimGym._global_step_counter == c0 + 1


def test_smoke_script_call_surface():
    """The calls of test_real_mappo.py with small budgets: import order, constructor, train_episode keys, the
    experiment entry point, evaluate(), save/load through the module-level names."""
    import MRRL
    from mappo_algorithm import MAPPO, RolloutBuffer  # noqa: F401
    from mappo_trainer import MAPPOTrainer, run_mappo_experiment  # noqa: F401
    cfg = dict(MRRL.ENV_CONFIG, episode_length=50)
    trainer = MAPPOTrainer(cfg, seed=0, verbose=False)
    stats = trainer.train_episode(explore=True)
    for k in ("episode_reward", "episode_steps", "avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps",
              "load_balance_jain"):
        assert k in stats
    assert stats["episode_steps"] == 50
    out = trainer.mappo.update()
    assert set(out) == {"actor_loss", "critic_loss", "entropy"}
    log = trainer.train(num_iterations=3, update_interval=1, eval_interval=2, eval_episodes=2, max_total_steps=150)
    assert len(log) == 1 and log[0]["iteration"] == 2 and log[0]["timestep"] == 100
    for k in ("algorithm", "seed", "episode_return_mean", "avg_latency_ms", "p99_latency_ms", "avg_energy_J",
              "throughput_tps", "load_balance_jain"):
        assert k in log[0]
    ev = trainer.evaluate(num_episodes=2)
    assert set(ev) == {"avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain"}
    assert MRRL.MAPPO_AVAILABLE and callable(MRRL.run_greedy_baseline)


def test_trainer_reports_exact_p99_at_4096_envs():
    """MAPPOTrainer at 4,096 environments: p99_latency_ms comes from the per-environment latency logs (no environment
    without one) and is the mean of the environments' exact percentiles, not 0.0."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(dict(ENV_CONFIG, episode_length=300), seed=2, device="cuda:0", n_envs=4096, verbose=False)
    st = tr.train_episode(explore=True, record_buffer=True)
    assert st["p99_envs_missing"] == 0 and st["p99_latency_ms"] > st["avg_latency_ms"] > 0
    import torch
    assert not bool(torch.isnan(st["per_env"]["metrics"]).any())
    ev = tr.evaluate(num_episodes=4096)
    assert ev["p99_latency_ms"] > ev["avg_latency_ms"] > 0
