"""End-to-end check of the scaled-topology trainers (marl_amd/ippo.py): rollout indexing, per-agent policy and critic,
action scatter into the environment's slot arrays, reward rows, per-agent GAE and the grouped update, replayed against
the numpy environment restatement and the torch-CPU restatement of the per-agent PPO arithmetic (oracle/ippo_oracle.py).

The environment side is bit-exact; policy outputs 1e-5 relative; parameters after the update 1e-4 relative / 2e-6
absolute (lr*m/(sqrt(v)+eps) amplifies last-bit gradient differences, as in tests/test_mappo_gpu.py).  Parity with the
reference itself is unpinned for independent PPO (RLlib not vendored)."""
import numpy as np
import pytest
import torch

from oracle import edge_env_oracle as eo
from oracle import ippo_oracle as io

pytestmark = pytest.mark.gpu


def _cfg(T):
    cfg = eo.default_config()
    cfg.update(n_nodes=6, n_deciders=14, node_cpu_capacity=np.linspace(1.0, 3.0, 6) * 1e9, task_arrival_rate_range=(50, 150),
               episode_length=T, disable_env_exploration=False)
    cfg["exploration_schedule"] = {"initial_epsilon": 0.5, "final_epsilon": 0.05, "decay_steps": 40}
    return cfg


def _params_close(got, want, lr=3e-4, steps=2):
    """1e-4 relative / 2e-6 absolute, except that a parameter whose gradient is ~1e-8 (Adam's eps) may move by a
    different fraction of lr per step: such outliers must stay below 1e-4 of the entries and within lr * steps."""
    d = np.abs(got - want)
    bad = d > 2e-6 + 1e-4 * np.abs(want)
    assert bad.mean() < 1e-4, bad.sum()
    assert d.max() <= lr * steps, d.max()


@pytest.mark.parametrize("shared", [False, True])
def test_episode_and_update_match_restatement(shared):
    from marl_amd.ippo import IPPOTrainer
    T, B, seed = 30, 5, 17
    cfg = _cfg(T)
    S, D = cfg["n_nodes"], cfg["n_deciders"]
    tr = IPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=B, shared=shared, verbose=False)
    tr.algo.num_sgd_iter = 2
    G = tr.algo.G
    o, n = tr.algo.actor.offsets["log_std"]
    tr.algo.actor.flat[:, o:o + n] = -0.5
    flat_a0, flat_c0 = tr.algo.actor.flat.cpu().clone(), tr.algo.critic.flat.cpu().clone()
    gen = torch.Generator().manual_seed(5)
    noise = torch.randn((tr.algo.buffer.capacity_entries, S + 1), generator=gen)
    tr.train_episode(explore=True, record_buffer=True, noise=noise.to("cuda:0").contiguous())
    ro = tr.algo.buffer
    E = ro.total()
    t = {k: v.cpu() for k, v in ro.t.items()}
    assert E > T          # the scenario produces work

    # ---- replay: numpy environments driven by the actions the device recorded ---------------------------------------
    envs = [eo.EdgeEnvOracle(cfg, eo.PhiloxDraws(seed, b, 1)) for b in range(B)]
    obs = [e.reset()[0] for e in envs]
    e_i = 0
    for step in range(T):
        for b, env in enumerate(envs):
            order = list(env.pending.keys())
            assert int(t["k_tb"][step, b]) == len(order) and (not order or int(t["first_tb"][step, b]) == e_i)
            pl, rs = np.full(D, -1, np.int64), np.zeros(D, np.float32)
            for k, d in enumerate(order):
                assert int(t["ent_env"][e_i]) == b and int(t["ent_agent"][e_i]) == d and int(t["ent_dst"][e_i]) == b * 64 + k
                assert np.array_equal(t["ent_obs"][e_i].numpy(), obs[b][d])
                p, r = io.format_actions(t["ent_action"][e_i:e_i + 1], S)
                pl[d], rs[d] = int(p[0]), float(r[0])
                e_i += 1
            out = env.step(pl, rs)
            obs[b] = out.obs
            assert float(t["reward_tb"][step, b]) == (out.reward if out.reward_emitted else 0.0)
            assert int(t["done_tb"][step, b]) == int(out.done)
    assert e_i == E

    # ---- policy and critic outputs of the recorded entries -------------------------------------------------------------
    agent = t["ent_agent"][:E].long() if not shared else torch.zeros(E, dtype=torch.int64)
    x = t["ent_obs"][:E]
    mu = io.forward(flat_a0, S + 4, S + 1, True, x, agent)
    ls = torch.full((E, S + 1), -0.5)
    a, lp = io.sample(mu, ls, noise[:E])
    np.testing.assert_allclose(t["ent_action"][:E].numpy(), a.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(t["ent_logp"][:E].numpy(), lp.numpy(), rtol=2e-5, atol=2e-4)
    v = io.forward(flat_c0, S + 4, 1, False, x, agent).squeeze(-1) * np.sqrt(1 + 1e-8)
    np.testing.assert_allclose(t["ent_value"][:E].numpy(), v.numpy(), rtol=1e-5, atol=1e-5)

    # ---- update: GAE chains, per-learner normalisation, two epochs ---------------------------------------------------------
    stats = tr.algo.update()
    adv, ret = io.gae_agents(t["reward_tb"].numpy(), t["done_tb"].numpy(), t["k_tb"].numpy(), t["first_tb"].numpy(),
                             t["ent_agent"].numpy(), t["ent_value"].numpy())
    assert np.array_equal(ro.t["ent_adv"][:E].cpu().numpy(), adv[:E]) and np.array_equal(ro.t["ent_ret"][:E].cpu().numpy(), ret[:E])
    fa, fc = flat_a0.clone(), flat_c0.clone()
    ma, va, mc, vc = (torch.zeros_like(fa), torch.zeros_like(fa), torch.zeros_like(fc), torch.zeros_like(fc))
    adv_t, ret_t = torch.as_tensor(adv[:E]), torch.as_tensor(ret[:E])
    sums = np.zeros(2)
    for step in (1, 2):
        ga, la = io.actor_grads(fa, S + 4, S + 1, x, t["ent_action"][:E], t["ent_logp"][:E], adv_t, agent)
        gc, lc = io.critic_grads(fc, S + 4, x, ret_t, agent)
        io.clip_adam(fa, ga, ma, va, step)
        io.clip_adam(fc, gc, mc, vc, step)
        sums += [la[:, 0].sum().item(), lc[:, 0].sum().item()]
    _params_close(tr.algo.actor.flat.cpu().numpy(), fa.numpy())
    _params_close(tr.algo.critic.flat.cpu().numpy(), fc.numpy())
    assert stats["transitions"] == E
    from tests.golden_util import assert_losses_close
    assert_losses_close([stats["actor_loss"], stats["critic_loss"], 1.0], [-sums[0] / (2 * E), 0.5 * sums[1] / (2 * E), 1.0],
                        "two epochs vs the torch restatement")


def test_scaled_64x256_smoke_and_eval():
    """The configs[4] topology end to end: one recorded episode + update, then a deterministic evaluation episode."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 25
    tr = IPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=96, verbose=False)
    before = tr.algo.actor.flat.clone()
    st = tr.train_episode()
    up = tr.algo.update()
    assert np.isfinite(st["episode_reward"]) and st["episode_reward"] < 0
    assert up["transitions"] > 96 * 25 and np.isfinite(up["actor_loss"]) and np.isfinite(up["critic_loss"])
    assert bool(torch.isfinite(tr.algo.actor.flat).all()) and not torch.equal(before, tr.algo.actor.flat)
    ev = tr.evaluate(num_episodes=96)
    assert all(np.isfinite(v) for v in ev.values())


def test_two_wavefront_grouped_backward_equals_the_one_wavefront_kernel(monkeypatch):
    """gmlp_bwd_kernel's SPLIT variant (dW1 and the tile staging in a second wavefront per SIMD) takes every partial sum in the
    order of the one-wavefront kernel: same episodes, same updates, same parameters bit for bit (many tiles per learner)."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 40
    for algorithm in ("ippo", "shared"):
        outs = []
        for split in ("1", "0"):
            monkeypatch.setenv("EM_BWD_SPLIT", split)
            tr = IPPOTrainer(cfg, seed=3, device="cuda:0", n_envs=2048, algorithm=algorithm, verbose=False)
            for _ in range(2):
                tr.train_episode(explore=True, record_buffer=True)
                up = tr.algo.update()
            outs.append((tr.algo.actor.flat.clone(), tr.algo.critic.flat.clone(), up))
        (aa, ca, ua), (ab, cb, ub) = outs
        assert np.isfinite(ua["critic_loss"]) and ua == ub, (ua, ub)
        assert torch.equal(aa, ab) and torch.equal(ca, cb)


def test_two_launch_index_equals_the_three_launch_path(monkeypatch):
    """ems_rollout_index with the scan folded into the placement launch (histograms and cursors alternating by step parity)
    records what the three-launch path records: same counters, entry numbers, group offsets and entry rows, the same SET of
    entries per group (the order inside a group comes from integer atomics in both), over two episodes in a row (odd
    horizon: the parity of the last step of an episode equals that of the next one's first) and the same parameters after
    the updates (the update sorts stably)."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 23
    outs = []
    for three in ("0", "1"):
        monkeypatch.setenv("EMS_INDEX_THREE_LAUNCHES", three)
        tr = IPPOTrainer(cfg, seed=5, device="cuda:0", n_envs=701, algorithm="ippo", verbose=False)
        rec = []
        for _ in range(2):
            tr.train_episode(explore=True, record_buffer=True)
            ro = tr.algo.buffer
            E, tt = ro.total(), ro.t
            rec.append((E, tt["counters"].clone(), tt["k_tb"].clone(), tt["first_tb"].clone(), tt["ent_obs"][:E].clone(),
                        tt["ent_agent"][:E].clone(), tt["ent_env"][:E].clone(), tt["group_off"].clone(), tt["perm"].clone()))
            up = tr.algo.update()
        outs.append((rec, tr.algo.actor.flat.clone(), tr.algo.critic.flat.clone(), up))
    (ra, aa, ca, ua), (rb, ab, cb, ub) = outs
    for x, y in zip(ra, rb):
        assert x[0] == y[0] and x[0] > 700
        for u, v in zip(x[1:8], y[1:8]):
            assert torch.equal(u, v)
        # the last step's permutation: the same entries in every group
        off = x[7].tolist()
        D = tr.algo.D
        for g in range(D):
            assert torch.equal(torch.sort(x[8][off[g]:off[g + 1]]).values, torch.sort(y[8][off[g]:off[g + 1]]).values)
    assert ua == ub and torch.equal(aa, ab) and torch.equal(ca, cb)


def test_policy_launch_copies_the_observation_rows():
    """Deferred critic: ems_rollout_index is called without rows and em_gmlp_policy_gather reads each row from the
    environment's pending list and stores the entry copy itself.  Same storage (rows, actions, log-probs), same environment
    actions, same parameters after the updates as the index launch's copy + em_gmlp_policy, for independent learners and
    for the shared actor with the central critic."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 21
    for algorithm in ("ippo", "mappo"):
        outs = []
        for gather in (True, False):
            tr = IPPOTrainer(cfg, seed=7, device="cuda:0", n_envs=333, algorithm=algorithm, verbose=False)
            assert tr.algo.defer_values
            tr.algo.gather_obs = gather
            rec = []
            for _ in range(2):
                st = tr.train_episode(explore=True, record_buffer=True)
                ro = tr.algo.buffer
                E, tt = ro.total(), ro.t
                rec.append((E, st["episode_reward"], tt["ent_obs"][:E].clone(), tt["ent_action"][:E].clone(), tt["ent_logp"][:E].clone(),
                            tt["ent_dst"][:E].clone()))
                up = tr.algo.update()
            outs.append((rec, tr.algo.actor.flat.clone(), tr.algo.critic.flat.clone(), up))
        (ra, aa, ca, ua), (rb, ab, cb, ub) = outs
        for x, y in zip(ra, rb):
            assert x[0] == y[0] and x[0] > 333 and x[1] == y[1]
            assert bool((x[2] != 0).any())
            for u, v in zip(x[2:], y[2:]):
                assert torch.equal(u, v)
        assert ua == ub and torch.equal(aa, ab) and torch.equal(ca, cb), algorithm


@pytest.mark.parametrize("three", ["0", "1"])
def test_scaled_storage_overflow_is_reported(monkeypatch, three):
    """A storage too small for the episode: the step that does not fit records nothing (its counter does not move, its
    per-environment counts are zero), the sticky flag is set, nothing is written past the capacity, and the trainer raises
    instead of updating - on the two-launch index path and on the three-launch one."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    monkeypatch.setenv("EMS_INDEX_THREE_LAUNCHES", three)
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 12
    B, cap = 128, 700                       # ~2.2 entries per environment and step: full after two or three steps
    tr = IPPOTrainer(cfg, seed=2, device="cuda:0", n_envs=B, algorithm="ippo", capacity_entries=cap, verbose=False)
    ro = tr.algo.buffer
    guard = ro.t["ent_env"]                  # [cap] int32: the kernels must not write behind it
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        tr.train_episode(explore=True, record_buffer=True)
        tr.algo.update()
    assert int(ro.t["err_flag"].item()) & 8
    c = ro.t["counters"].cpu().tolist()
    k = ro.t["k_tb"].cpu().numpy().astype(np.int64).sum(axis=1)
    assert c[0] == 0 and max(c) <= cap and all(b >= a for a, b in zip(c, c[1:]))
    moved = [b - a for a, b in zip(c, c[1:])]
    assert [int(v) for v in k] == moved[:len(k)], (list(k), moved)
    assert 0 in moved[2:] and max(moved) > B
    E = c[-1]
    assert bool((guard[:E] >= 0).all()) and bool((guard[:E] < B).all())
    first = ro.t["first_tb"].cpu().numpy()
    assert first.max() <= cap
