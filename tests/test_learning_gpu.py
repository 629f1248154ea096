"""Behavioural evidence that the trainers LEARN (last row: the substitute for the un-pinnable RLlib parity of
IPPO, and a sanity check of MAPPO beside its pinned arithmetic): with a fixed seed the mean training reward of the last
updates beats the first by a stated margin.  The environment's own epsilon-greedy override (MRRL.PY:258-274) is switched
off so that the training reward reflects the policy; everything else is the reference's hyper-parameters (lr 3e-4, 4
full-batch epochs per update).  Margins come from tools/learning_curve.py (MI355X, seeds 0 and 1: MAPPO -295 -> -245,
IPPO -411 -> -344 over 120 updates; about 60 % of that after 60)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ITERS, T, ENVS = 60, 200, 256
NO_EPS = {"initial_epsilon": 0.0, "final_epsilon": 0.0, "decay_steps": 1}


def _improved(curve, margin):
    c = np.asarray(curve, np.float64)
    first, last = c[:5].mean(), c[-10:].mean()
    assert np.isfinite(c).all()
    assert last > first + margin * abs(first), (first, last, c.tolist())
    # not a fluke of the end points: the middle of the run already sits above the start
    assert c[ITERS // 2 - 5:ITERS // 2 + 5].mean() > first


def test_mappo_training_reward_improves():
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=T, disable_env_exploration=True)
    tr = MAPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=ENVS, verbose=False)
    curve = []
    for _ in range(ITERS):
        curve.append(tr.train_episode(explore=True, record_buffer=True)["episode_reward"])
        out = tr.mappo.update()
        assert np.isfinite(out["critic_loss"])
    _improved(curve, 0.08)
    tr.env.close()


def test_ippo_through_run_experiment_improves():
    """``MRRL.run_experiment("IPPO", ...)`` (MRRL.PY:573-638) on a small scaled topology (8 servers x 24 independent
    learners): the log has the reference's keys, evaluation metrics are attached at the evaluation interval, and the
    training reward improves."""
    import MRRL
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG, n_nodes=8, n_deciders=24, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9, episode_length=T,
               task_arrival_rate=20, task_arrival_rate_range=(10, 30), exploration_schedule=NO_EPS)
    log = MRRL.run_experiment("IPPO", seed=0, num_iterations=3, max_timesteps=T * ITERS, eval_interval_steps=T * 20, n_envs=ENVS,
                              env_config=cfg, eval_episodes=ENVS, verbose=False)
    assert len(log) == ITERS                       # the step budget rules; num_iterations=3 is raised like the reference's cap
    assert [e["timestep"] for e in log] == [T * (i + 1) for i in range(ITERS)]
    assert all(e["algorithm"] == "IPPO" and e["seed"] == 0 for e in log)
    with_eval = [i for i, e in enumerate(log) if "avg_latency_ms" in e]
    assert with_eval == [19, 39, 59]
    for i in with_eval:
        assert all(np.isfinite(log[i][k]) for k in ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain"))
    _improved([e["episode_return_mean"] for e in log], 0.08)
    # the deterministic policy got better too: evaluation latency falls between the first and the last evaluation
    assert log[59]["avg_latency_ms"] < log[19]["avg_latency_ms"]


def test_run_experiment_dispatch(tmp_path, monkeypatch):
    import MRRL
    monkeypatch.chdir(tmp_path)                    # run_mappo_experiment saves mappo_seed_<seed>.pt like the reference (:391)
    log = MRRL.run_experiment("MAPPO", seed=2, num_iterations=1, max_timesteps=1000, n_envs=4)
    assert log == [] and (tmp_path / "mappo_seed_2.pt").exists()      # one iteration: no evaluation entry yet (eval every 10)
    for name in ("MADDPG", "QMIX"):
        with pytest.raises(NotImplementedError):
            MRRL.run_experiment(name, seed=0)
    with pytest.raises(ValueError):
        MRRL.run_experiment("PPO2", seed=0)
    g = MRRL.run_experiment("Greedy", seed=0, num_iterations=3)
    assert g[0]["algorithm"] == "Greedy" and np.isfinite(g[0]["avg_latency_ms_mean"])
