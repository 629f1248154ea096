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

        raws = np.zeros((T, n, n + 1), np.float32)
        od, _ = env.reset(seed=seed)
        put(0, od)
        load[0], queue[0], sim_time[0] = env.nodes_load, env.nodes_queue_len, env.sim_time
        metrics = None
        for t in range(T):
            if action_mode == "raw":
                # continuous action vectors go through the reference's own _convert_action (MRRL.PY:276-305)
                raw = act_rs.normal(0.0, 1.2, size=(n, n + 1)).astype(np.float32)
                for a in range(n):
                    c = act_rs.randint(0, 12)
                    if c == 0:
                        raw[a, :n] = 0.0                       # allclose(logits, 0) -> placement 0
                    elif c == 1:
                        raw[a, :n] = act_rs.uniform(-9e-9, 9e-9, size=n).astype(np.float32)
                    elif c == 2:
                        raw[a, act_rs.randint(0, n)] = raw[a, :n].max()   # tie: first maximum wins
                    elif c == 3:
                        raw[a, n] = np.float32(act_rs.choice([-1.0, 1.0, -3.5, 2.25, 0.0]))
                raws[t] = raw
                actions = {k: raw[int(k.split("_")[1])].copy() for k in od}
                conv = [env._convert_action(raw[a]) for a in range(n)]
                pl = np.array([c["placement"] for c in conv], np.int32)
                rsrc = np.array([c["resource"][0] for c in conv], np.float32)
            elif action_mode == "greedy":
                # the reference's greedy baseline policy, verbatim in behaviour (MRRL.PY:803-812)
                target = int(np.argmin(env.nodes_load / env.cpu_caps))
                pl, rsrc = np.full(n, target, np.int32), np.full(n, 0.5, np.float32)
                actions = {k: {"placement": target, "resource": np.array([0.5], dtype=np.float32)} for k in od}
            else:
                pl, rsrc = policy_actions(act_rs, n, action_mode)
                actions = {k: {"placement": int(pl[int(k.split("_")[1])]),
                               "resource": np.array([rsrc[int(k.split("_")[1])]], dtype=np.float32)} for k in od}
            placements[t], resources[t] = pl, rsrc
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
        placements=placements, resources=resources, raws=raws, action_mode=action_mode,
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
    ("seed6_rawvec_T150", 6, {"episode_length": 150, "disable_env_exploration": True}, "raw", 0),
    ("seed7_greedy_T150", 7, {"episode_length": 150, "disable_env_exploration": True}, "greedy", 0),
]


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    for sc in SCENARIOS:
        if not only or sc[0] in only:
            run_scenario(*sc)




# ------------------------------------------------------------------------------------------------
# MAPPO goldens: one reference episode of MAPPOTrainer(seed=0) + the first update()
# ------------------------------------------------------------------------------------------------
HYPER = {"lr_act": 1e-4, "lr_cri": 1e-3, "clip": 0.15, "entropy": 0.005, "num_sgd": 6, "batch": 256}


def run_mappo_scenario(name="mappo_seed0_T200", seed=0, T=200):
    import copy

    import torch

    MRRL, alg, trainer_mod = load_reference()
    real_np = np
    proxy = NumpyProxy(real_np)
    eps_log = []

    BaseNormal = alg.Normal

    class RecordingNormal(BaseNormal):
        """Normal whose rsample logs the standard-normal draw (same formula as torch's rsample)."""

        def rsample(self, sample_shape=torch.Size()):
            shape = self._extended_shape(sample_shape)
            eps = torch.normal(torch.zeros(shape, dtype=self.loc.dtype), torch.ones(shape, dtype=self.loc.dtype))
            eps_log.append(eps.detach().numpy().copy().reshape(-1))
            return self.loc + eps * self.scale

    MRRL.np = proxy
    alg.Normal = RecordingNormal
    try:
        cfg = {**MRRL.ENV_CONFIG, "episode_length": T}
        MRRL.EdgeSimGym._global_step_counter = 0
        tr = trainer_mod.MAPPOTrainer(cfg, seed=seed, device="cpu")
        n = cfg["n_nodes"]
        init_params = {f"actor.{k}": v.detach().numpy().copy() for k, v in tr.mappo.actor.state_dict().items()}
        init_params.update({f"critic.{k}": v.detach().numpy().copy() for k, v in tr.mappo.critic.state_dict().items()})
        proxy.random.log.clear()
        env_actions = []
        orig_step = tr.env.step

        def spy_step(actions):
            pl = np.full(n, -1, np.int32)
            rs = np.zeros(n, np.float32)
            for k, a in actions.items():
                i = int(k.split("_")[1])
                pl[i] = a["placement"]
                rs[i] = a["resource"][0]
            env_actions.append((pl, rs))
            return orig_step(actions)

        tr.env.step = spy_step
        counter0 = MRRL.EdgeSimGym._global_step_counter
        stats = tr.train_episode(explore=True, record_buffer=True)
        flat = list(proxy.random.log)
        buf = tr.mappo.buffer.get()
        L = len(buf["rewards"])
        b_obs = np.array(buf["observations"], np.float32)
        b_states = np.array(buf["states"], np.float32)
        b_actions = np.array(buf["actions"], np.float32)
        b_rewards = np.array(buf["rewards"], np.float64)
        b_dones = np.array(buf["dones"], np.uint8)
        b_logp = np.array(buf["log_probs"], np.float64)
        b_values = np.array(buf["values"], np.float64)
        adv, ret = tr.mappo.compute_gae(buf["rewards"], buf["values"], buf["dones"], 0)

        m1 = copy.deepcopy(tr.mappo)
        m1.num_sgd_iter = 1
        loss1 = m1.update()
        params1 = {f"actor.{k}": v.detach().numpy().copy() for k, v in m1.actor.state_dict().items()}
        params1.update({f"critic.{k}": v.detach().numpy().copy() for k, v in m1.critic.state_dict().items()})
        # run_mappo_batch.py:255-261: the sweep pokes these attributes after construction; one update with every one of
        # them off its default (distinct learning rates per network, 6 epochs) pins that they take effect
        mh = copy.deepcopy(tr.mappo)
        mh.actor_optimizer.param_groups[0]["lr"] = HYPER["lr_act"]
        mh.critic_optimizer.param_groups[0]["lr"] = HYPER["lr_cri"]
        mh.clip_param, mh.entropy_coeff, mh.num_sgd_iter = HYPER["clip"], HYPER["entropy"], HYPER["num_sgd"]
        mh.mini_batch_size = HYPER["batch"]
        lossh = mh.update()
        paramsh = {f"actor.{k}": v.detach().numpy().copy() for k, v in mh.actor.state_dict().items()}
        paramsh.update({f"critic.{k}": v.detach().numpy().copy() for k, v in mh.critic.state_dict().items()})
        loss4 = tr.mappo.update()
        params4 = {f"actor.{k}": v.detach().numpy().copy() for k, v in tr.mappo.actor.state_dict().items()}
        params4.update({f"critic.{k}": v.detach().numpy().copy() for k, v in tr.mappo.critic.state_dict().items()})
    finally:
        MRRL.np = real_np
        alg.Normal = BaseNormal

    # structured env tape via the restatement (also re-checks draw order)
    tape = eo.empty_tape(T + 1, n, a_slots=32)
    ocfg = {**eo.default_config(), "episode_length": T}
    orc = eo.EdgeEnvOracle(ocfg, eo.FlatReplay(flat), tape_out=tape)
    orc.explore_counter = counter0
    obs0, order = orc.reset()
    step_obs = [obs0]
    step_rank = [np.array([order.index(a) if a in order else -1 for a in range(n)], np.int8)]
    step_reward, step_emitted = [], []
    for t in range(T):
        pl, rs = env_actions[t]
        out = orc.step(np.where(pl < 0, 0, pl), rs, present={a for a in range(n) if pl[a] >= 0})
        step_obs.append(out.obs)
        step_rank.append(np.array([out.order.index(a) if a in out.order else -1 for a in range(n)], np.int8))
        step_reward.append(out.reward)
        step_emitted.append(out.reward_emitted)
    assert orc.draws.pos == len(flat)
    # the buffer must be the per-agent flattening of these steps (mappo_trainer.py:160-175)
    i = 0
    for t in range(T):
        rk = step_rank[t]
        for r in range(int((rk >= 0).sum())):
            a = int(np.where(rk == r)[0][0])
            assert np.array_equal(b_obs[i], step_obs[t][a]) and np.array_equal(b_states[i], step_obs[t].reshape(-1))
            assert b_rewards[i] == (step_reward[t] if step_emitted[t] else 0.0)
            i += 1
    assert i == L
    eps = np.concatenate(eps_log).reshape(L, n + 1).astype(np.float32)
    out_path = os.path.join(GOLDEN, f"{name}.npz")
    np.savez_compressed(
        out_path, seed=seed, episode_length=T, n_nodes=n, explore_counter0=counter0,
        episode_reward=stats["episode_reward"],
        env_placement=np.array([a[0] for a in env_actions]), env_resource=np.array([a[1] for a in env_actions]),
        step_obs=np.array(step_obs), step_rank=np.array(step_rank), step_reward=np.array(step_reward),
        step_emitted=np.array(step_emitted),
        buf_actions=b_actions, buf_logp=b_logp, buf_values=b_values, buf_dones=b_dones, buf_rewards=b_rewards,
        noise=eps, gae_adv=np.array(adv, np.float64), gae_ret=np.array(ret, np.float64),
        loss1=np.array([loss1["actor_loss"], loss1["critic_loss"], loss1["entropy"]]),
        loss4=np.array([loss4["actor_loss"], loss4["critic_loss"], loss4["entropy"]]),
        lossh=np.array([lossh["actor_loss"], lossh["critic_loss"], lossh["entropy"]]),
        hyper=np.array([HYPER[k] for k in ("lr_act", "lr_cri", "clip", "entropy", "num_sgd")], np.float64),
        **{f"ph.{k}": v for k, v in paramsh.items()},
        **{f"init.{k}": v for k, v in init_params.items()},
        **{f"p1.{k}": v for k, v in params1.items()},
        **{f"p4.{k}": v for k, v in params4.items()},
        **{f"tape_{k}": v for k, v in tape.items()},
    )
    print(f"[golden] {out_path}: L={L}, episode_reward={stats['episode_reward']!r}, loss1={loss1}, loss4={loss4}")


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if not a.startswith("--")]
    if "--mappo" not in sys.argv:
        main()
    if "--mappo-full" in sys.argv:       # configs[0] at its real length (reward -1787.630, 883 transitions)
        run_mappo_scenario("mappo_seed0_T1000", 0, 1000)
    elif not names:
        run_mappo_scenario()
