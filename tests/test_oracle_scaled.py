"""CPU checks of the test infrastructure for the scaled topology: the generalised numpy restatement (n_deciders) and the
torch restatement of per-agent PPO.  (The reference pins the D == S case through tests/test_oracle_env.py.)"""
import numpy as np
import torch

from oracle import edge_env_oracle as eo
from oracle import ippo_oracle as io


def _cfg(**kw):
    cfg = eo.default_config()
    cfg.update(n_nodes=4, n_deciders=12, node_cpu_capacity=np.array([1.0, 1.7, 2.2, 3.0]) * 1e9,
               task_arrival_rate_range=(200, 600), episode_length=60, disable_env_exploration=True)
    cfg.update(kw)
    return cfg


def test_d_equals_s_is_the_reference_restatement():
    """n_deciders == n_nodes must change nothing: same draws, same outputs as the config without the key."""
    base = eo.default_config()
    base.update(episode_length=80)
    a = eo.EdgeEnvOracle(dict(base), eo.LegacyDraws(3))
    b = eo.EdgeEnvOracle(dict(base, n_deciders=10), eo.LegacyDraws(3))
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa[0], ob[0]) and oa[1] == ob[1]
    rs = np.random.RandomState(0)
    for _ in range(80):
        pl, r = rs.randint(0, 10, size=10), rs.uniform(0.1, 1, size=10).astype(np.float32)
        x, y = a.step(pl, r), b.step(pl, r)
        assert np.array_equal(x.obs, y.obs) and x.reward == y.reward and x.order == y.order
    assert a.episode_metrics() == b.episode_metrics()


def test_shared_home_servers_semantics():
    """D > S: a decider's observation is taken from its home server d mod S, duplicate arrivals only ever ADD to a
    home server's queue, and every pending decider is dispatched exactly once."""
    cfg = _cfg()
    env = eo.EdgeEnvOracle(cfg, eo.PhiloxDraws(11, 0, 1))
    obs, order = env.reset()
    assert obs.shape == (12, 8) and all(0 <= d < 12 for d in order)
    rs = np.random.RandomState(1)
    bumped, dispatched = 0, 0
    for _ in range(60):
        q_before = env.queue.copy()
        pending = list(env.pending)
        pl = rs.randint(0, 4, size=12)
        res = rs.uniform(0.1, 1, size=12).astype(np.float32)
        out = env.step(pl, res)
        assert len(out.latencies) == len(pending) and out.targets == [int(pl[d]) for d in pending]
        dispatched += len(pending)
        placed = np.bincount([int(pl[d]) for d in pending], minlength=4)
        floor = np.maximum(0, q_before + placed - 1)          # placements, then the decrement, no duplicate arrivals
        assert np.all(env.queue >= floor)
        bumped += int((env.queue > floor).sum())
        for d in out.order:                                    # zero rows for deciders without a task, width S + 4
            assert out.obs[d].shape == (8,) and out.obs[d][3] >= 1.0      # workload column of a real task
        assert not out.obs[[d for d in range(12) if d not in out.order]].any()
    assert bumped > 0 and dispatched == len(env.ep_lat)
    m = env.episode_metrics()
    assert 0.0 <= m["violation_rate"] <= 1.0 and m["load_balance_jain"] > 0.5


def test_gae_of_one_agent_matches_the_reference_recursion():
    """One agent, one environment: the per-agent chain is MAPPO.compute_gae's known answer."""
    k = np.ones((3, 1), np.uint8)
    first = np.arange(3, dtype=np.int32).reshape(3, 1)
    adv, ret = io.gae_agents(np.array([[1.0], [2.0], [3.0]]), np.array([[0], [0], [1]], np.uint8), k, first,
                             np.zeros(3, np.int32), np.array([0.5, 0.4, 0.3], np.float32))
    want = np.array([5.068387175000001, 4.436350000000001, 2.7])
    np.testing.assert_allclose(adv, want.astype(np.float32), rtol=1e-6)
    np.testing.assert_allclose(ret, (want + np.array([0.5, 0.4, 0.3])).astype(np.float32), rtol=1e-6)


def test_agents_do_not_chain_into_each_other():
    k = np.array([[2], [1], [1]], np.uint8)
    first = np.array([[0], [2], [3]], np.int32)
    agent = np.array([0, 1, 0, 1], np.int32)
    value = np.array([1.0, 2.0, 3.0, 4.0], np.float32)
    reward = np.array([[1.0], [1.0], [1.0]])
    done = np.zeros((3, 1), np.uint8)
    adv, _ = io.gae_agents(reward, done, k, first, agent, value)
    # agent 0: entries 0 (t=0) and 2 (t=1); agent 1: entries 1 (t=0) and 3 (t=2)
    g2 = 1.0 - 3.0
    g0 = (1.0 + 0.99 * 3.0 - 1.0) + 0.99 * 0.95 * g2
    g3 = 1.0 - 4.0
    g1 = (1.0 + 0.99 * 4.0 - 2.0) + 0.99 * 0.95 * g3
    np.testing.assert_allclose(adv, np.array([g0, g1, g2, g3], np.float32), rtol=1e-6)


def test_single_group_equals_plain_ppo_losses():
    """With one group the grouped restatement is the plain shared-actor objective (autograd check of its pieces)."""
    torch.manual_seed(0)
    in_dim, out_dim, rows = 9, 5, 40
    n_per = 64 * in_dim + 64 + 64 * 64 + 64 + out_dim * 64 + out_dim + out_dim
    flat = torch.randn(1, n_per) * 0.1
    x, noise = torch.randn(rows, in_dim), torch.randn(rows, out_dim)
    agent = torch.zeros(rows, dtype=torch.int64)
    mu = io.forward(flat, in_dim, out_dim, True, x, agent)
    ls = io.unpack(flat[0], in_dim, out_dim, True)["log_std"].expand(rows, out_dim)
    a, lp = io.sample(mu, ls, noise)
    dist = torch.distributions.Normal(mu, torch.exp(ls))
    np.testing.assert_allclose(lp.numpy(), dist.log_prob(a).sum(-1).numpy(), rtol=1e-5, atol=1e-5)
    g, l3 = io.actor_grads(flat, in_dim, out_dim, x, a, lp.detach(), torch.randn(rows), agent)
    assert l3[0, 2] == rows and torch.isfinite(g).all() and g.abs().sum() > 0
    np.testing.assert_allclose(l3[0, 1].item(), dist.entropy().sum().item(), rtol=1e-5)
