"""GPU parity of the scaled-topology step (ems_*, D deciders over S servers; BASELINE.json configs[4]).

The generalisation is build-defined (the reference has deciders == servers only), so the chain of evidence is:
 (1) the numpy restatement with ``n_deciders`` is the reference-pinned restatement when D == S (tests/test_oracle_env.py
     runs it against the reference goldens) - and the HIP scaled kernel at D == S == 10 equals the reference-pinned
     em_step kernel bit for bit on the same Philox stream;
 (2) at D != S the HIP kernel equals the numpy restatement bit for bit (64 servers x 256 deciders x 10x rate, shared
     home servers with duplicate arrivals, more than 64 arrivals per step);
 (3) size-independent properties at 4,096 environments.
"""
import numpy as np
import pytest
import torch

from oracle import edge_env_oracle as eo

pytestmark = pytest.mark.gpu

SEED = 0xBADC0DE77


def scaled_cfg(**kw):
    cfg = eo.default_config()
    cfg.update(n_nodes=64, n_deciders=256, node_cpu_capacity=np.linspace(1.0, 3.0, 64) * 1e9,
               task_arrival_rate=100, task_arrival_rate_range=(50, 150))
    cfg.update(kw)
    return cfg


def _oracles(cfg, B, off, episode=1):
    return [eo.EdgeEnvOracle(cfg, eo.PhiloxDraws(SEED, off + b, episode)) for b in range(B)]


def _check_lists(env, refs, outs, t):
    n_p = env.n_pending.cpu().numpy()
    dec = env.pend_decider.cpu().numpy()
    obs = env.pend_obs.cpu().numpy()
    for b, (r, o) in enumerate(zip(refs, outs)):
        order = list(r.pending.keys())
        assert n_p[b] == len(order), (t, b)
        assert list(dec[b, :len(order)]) == order, (t, b)
        r_obs = o if isinstance(o, np.ndarray) else o.obs
        for k, d in enumerate(order):
            assert np.array_equal(obs[b, k], r_obs[d]), (t, b, k, d)


def _run_against_numpy(cfg, B, T, explore, by_slot=False, episodes=1, epw=0):
    from marl_amd.env_scaled import ScaledEdgeSimGym, PENDING_CAP
    cfg = dict(cfg)
    cfg.update(episode_length=T, disable_env_exploration=not explore)
    cfg["exploration_schedule"] = {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 60}
    S, D, off = cfg["n_nodes"], cfg["n_deciders"], 700
    env = ScaledEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=SEED, env_offset=off, latency_log_capacity=64 * T)
    env.enable_debug_outputs()
    env.set_envs_per_wave(epw or 4)     # small test batches would pick one environment per wavefront automatically
    counter = 0
    for episode in range(1, episodes + 1):
        refs = _oracles(cfg, B, off, episode)
        for r in refs:
            r.explore_counter = counter
        env.reset()
        _check_lists(env, refs, [r.reset()[0] for r in refs], -1)
        rs = np.random.RandomState(5 + episode)
        for t in range(T):
            pl = rs.randint(0, S, size=(B, D)).astype(np.int32)
            rsrc = rs.uniform(0.1, 1.0, size=(B, D)).astype(np.float32)
            outs = [r.step(pl[b], rsrc[b]) for b, r in enumerate(refs)]
            if by_slot:   # the same actions, aligned with the pending list the previous call returned
                dec = env.pend_decider.cpu().numpy().astype(np.int64) % D
                pl_s = np.take_along_axis(pl, dec, axis=1).astype(np.int32)
                rs_s = np.take_along_axis(rsrc, dec, axis=1).astype(np.float32)
                assert pl_s.shape == (B, PENDING_CAP)
                env.step(torch.as_tensor(pl_s).to(env.device), torch.as_tensor(rs_s).to(env.device), by_slot=True)
            else:
                env.step(torch.as_tensor(pl).to(env.device), torch.as_tensor(rsrc).to(env.device))
            _check_lists(env, refs, outs, t)
            assert list(env.reward_raw.cpu().numpy()) == [o.reward for o in outs], t
            assert list(env.emitted.cpu().numpy()) == [int(o.reward_emitted) for o in outs], t
            assert list(env.reward.cpu().numpy()) == [o.reward if o.reward_emitted else 0.0 for o in outs], t
            assert list(env.done.cpu().numpy()) == [int(o.done) for o in outs], t
            sl = env.step_latency.cpu().numpy()
            for b, o in enumerate(outs):
                k = len(o.latencies)
                assert list(sl[b, :k]) == o.latencies and np.all(np.isnan(sl[b, k:])), (t, b)
        st = env.get_state()
        for b, r in enumerate(refs):
            assert np.array_equal(st["load"][b], r.load) and np.array_equal(st["queue"][b], r.queue), b
            assert st["sim_time"][b] == r.sim_time and st["elapsed"][b] == r.elapsed and st["step"][b] == r.step_no
            assert list(st["counts"][b]) == [len(r.ep_lat), r.ep_viol, r.ep_done_tasks]
        m = env.episode_metrics().cpu().numpy()
        for b, r in enumerate(refs):
            rm = r.episode_metrics()
            want = np.array([rm[k] for k in ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "violation_rate",
                                             "throughput_tps", "load_balance_jain")])
            assert m[b, 1] == want[1] and m[b, 3] == want[3] and m[b, 4] == want[4], b
            np.testing.assert_allclose(m[b], want, rtol=1e-12, atol=0)
        if explore:
            counter += T
    assert env.global_step_counter == counter
    env.close()


@pytest.mark.parametrize("epw", [4, 1])
@pytest.mark.parametrize("explore", [False, True])
def test_scaled_64x256_matches_numpy_restatement(explore, epw):
    _run_against_numpy(scaled_cfg(), B=9, T=120, explore=explore, episodes=2 if explore else 1, epw=epw)


def test_scaled_actions_by_slot():
    _run_against_numpy(scaled_cfg(), B=5, T=60, explore=True, by_slot=True)


def test_shared_home_servers_and_duplicate_arrivals():
    """12 deciders on 4 servers at a high rate: most arrivals hit a busy decider or a home server whose queue another
    decider's duplicate just bumped (the case D == S never produces)."""
    cfg = scaled_cfg(n_nodes=4, n_deciders=12, node_cpu_capacity=np.array([1.0, 1.7, 2.2, 3.0]) * 1e9,
                     task_arrival_rate_range=(200, 600), task_arrival_mode="random")
    _run_against_numpy(cfg, B=7, T=150, explore=False)      # mixes 16-lane groups and the general path inside a wavefront
    _run_against_numpy(cfg, B=7, T=60, explore=True, epw=1)


def test_more_than_64_arrivals_per_step():
    """window fixed at 0.2 s, 400-500 arrivals/s: 80-100 arrivals per step over 40 deciders (sequential tail)."""
    cfg = scaled_cfg(n_nodes=8, n_deciders=40, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9,
                     task_arrival_rate_range=(400, 500), min_time_step=0.2, max_time_step=0.2,
                     initial_observation_window=0.2)
    _run_against_numpy(cfg, B=3, T=40, explore=True)


@pytest.mark.parametrize("explore", [False, True])
def test_reduces_to_the_reference_step_at_d_equals_s(explore):
    """D == S == 10 with the identity home map: the scaled kernel must reproduce em_step (pinned to the reference's
    goldens by tests/test_env_gpu.py) bit for bit on the same Philox stream."""
    from marl_amd.env import BatchedEdgeSimGym
    from marl_amd.env_scaled import ScaledEdgeSimGym
    T, B, n = 200, 70, 10
    cfg = eo.default_config()
    cfg.update(episode_length=T, disable_env_exploration=not explore)
    cfg["exploration_schedule"] = {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 150}
    a = BatchedEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=SEED, env_offset=31, latency_log_capacity=T * n)
    b = ScaledEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=SEED, env_offset=31, latency_log_capacity=T * n)
    b.set_envs_per_wave(4 if explore else 1)
    a.reset()
    b.reset()
    assert torch.equal(a.obs, b.dense_obs())
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    for t in range(T):
        pl = torch.randint(0, n, (B, n), device="cuda:0", generator=gen, dtype=torch.int32)
        rs = (torch.rand((B, n), device="cuda:0", generator=gen) * 0.9 + 0.1).contiguous()
        a.step(pl, rs)
        b.step(pl, rs)
        assert torch.equal(a.obs, b.dense_obs()), t
        assert torch.equal(a.reward_raw, b.reward_raw) and torch.equal(a.reward, b.reward), t
        assert torch.equal(a.emitted, b.emitted) and torch.equal(a.done, b.done), t
        # arrival ranks: slot k of the pending list is the agent with rank k
        rank = a.rank.cpu().numpy()
        dec = b.pend_decider.cpu().numpy()
        for e in range(B):
            order = [int(np.where(rank[e] == k)[0][0]) for k in range(int((rank[e] >= 0).sum()))]
            assert list(dec[e, :len(order)]) == order, (t, e)
    sa, sb = a.get_state(), b.get_state()
    for k in ("load", "queue", "sim_time", "elapsed", "step", "counts"):
        assert np.array_equal(sa[k], sb[k]), k
    assert torch.equal(a.episode_metrics(), b.episode_metrics())
    a.close()
    b.close()


def test_properties_at_4096_envs():
    """Full batch: shard invariance (two handles over halves of the global environment range replay the union handle),
    task conservation, clock bounds, done flag at the horizon."""
    from marl_amd.env_scaled import ScaledEdgeSimGym
    cfg = scaled_cfg(episode_length=100, disable_env_exploration=False)
    B, S, D, T = 4096, 64, 256, 100
    whole = ScaledEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=9, env_offset=0)
    lo = ScaledEdgeSimGym(cfg, n_envs=B // 2, device="cuda:0", seed=9, env_offset=0)
    hi = ScaledEdgeSimGym(cfg, n_envs=B // 2, device="cuda:0", seed=9, env_offset=B // 2)
    whole.set_envs_per_wave(4)          # the union on four environments per wavefront, the halves on one
    for e in (whole, lo, hi):
        e.reset()
    gen = torch.Generator(device="cuda:0").manual_seed(1)
    acted = torch.zeros(B, dtype=torch.int64, device="cuda:0")
    for t in range(T):
        pl = torch.randint(0, S, (B, D), device="cuda:0", generator=gen, dtype=torch.int32)
        rs = (torch.rand((B, D), device="cuda:0", generator=gen) * 0.9 + 0.1).contiguous()
        acted += whole.n_pending.long()
        whole.step(pl, rs)
        lo.step(pl[:B // 2].contiguous(), rs[:B // 2].contiguous())
        hi.step(pl[B // 2:].contiguous(), rs[B // 2:].contiguous())
        assert torch.equal(whole.n_pending, torch.cat([lo.n_pending, hi.n_pending])), t
        assert torch.equal(whole.reward, torch.cat([lo.reward, hi.reward])), t
        assert bool((whole.done == (1 if t == T - 1 else 0)).all())
    live = torch.arange(64, device="cuda:0")[None, :] < whole.n_pending[:, None]
    assert torch.equal(whole.pend_obs[live], torch.cat([lo.pend_obs, hi.pend_obs])[live])
    st = whole.get_state()
    assert np.array_equal(st["counts"][:, 0], acted.cpu().numpy())                 # every pending task was dispatched
    assert np.array_equal(st["counts"][:, 1] + st["counts"][:, 2], st["counts"][:, 0])   # violated + completed = tasks
    assert np.all(st["sim_time"] >= 0.02 * T - 1e-9) and np.all(st["sim_time"] <= 0.2 * T + 1e-9)
    assert np.all(st["load"] >= 0) and np.all(st["queue"] >= 0)
    assert 1.5 < acted.float().mean().item() / T < 3.5    # ~ Poisson(rate * 0.02) distinct deciders per step
    for e in (whole, lo, hi):
        e.close()
