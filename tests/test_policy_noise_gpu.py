"""The in-kernel policy-noise stream (em_policy_step / em_gmlp_policy with noise == NULL: every throughput run) against
its restatement oracle/philox_ref.policy_noise_np: the counterpart of Normal(mu, exp(log_std)).rsample() and
.log_prob().sum(-1) in MAPPO.get_actions (mappo_algorithm.py:206-213, 62-71).

  * stored action == mu + sigma * eps_restated     (abs 5e-6 + rel 2e-6: float32 rounding of logf/sincosf, which are
                                                    not bit-reproducible on the host; mu comes from em_mlp3_forward,
                                                    itself held to fp32 torch in tests/test_mlp_gpu.py)
  * stored log-prob == log N(a; mu, sigma)         (1e-5 relative, recomputed in float64 from the stored action)
  * eps ~ N(0, 1): moments and a Kolmogorov-Smirnov test over more than a million draws
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HALF_LOG_2PI = 0.9189385332046727


def _entry_steps(counters, E):
    """Step index of every entry from the per-step entry bases (counters[t][0])."""
    base = counters[:, 0].cpu().numpy()
    return np.searchsorted(base, np.arange(E), side="right") - 1


def _check(action, logp, mu, log_std, eps, label):
    sigma = np.exp(log_std.astype(np.float32)).astype(np.float32)
    want = mu + sigma[None, :] * eps
    err = np.abs(action - want)
    tol = 5e-6 + 2e-6 * np.abs(want)
    print(f"{label}: {action.shape[0]} entries x {action.shape[1]} components; max |a - (mu + sigma eps)| = {err.max():.3e}")
    assert (err <= tol).all(), f"{label}: sampled actions leave the restated noise stream by {err.max():.3e}"
    a64, m64, s64 = action.astype(np.float64), mu.astype(np.float64), sigma.astype(np.float64)
    lp = (-(a64 - m64) ** 2 / (2.0 * s64 ** 2) - log_std.astype(np.float64) - HALF_LOG_2PI).sum(axis=1)
    np.testing.assert_allclose(logp, lp, rtol=1e-5, atol=1e-5, err_msg=f"{label}: stored log-prob")
    # distribution of the draws the kernel used (implied by its own outputs)
    z = ((a64 - m64) / s64).ravel()
    n = z.size
    assert abs(z.mean()) < 5.0 / np.sqrt(n) and abs(z.var() - 1.0) < 5.0 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 5.0 * np.sqrt(15.0 / n) and abs((z ** 4).mean() - 3.0) < 5.0 * np.sqrt(96.0 / n)
    from scipy import stats
    sub = z[:: max(1, n // 200_000)]
    assert stats.kstest(sub, "norm").pvalue > 1e-4
    # components of one row are independent draws: no correlation between neighbours
    zz = ((a64 - m64) / s64)
    c = np.corrcoef(zz[:, 0], zz[:, 1])[0, 1]
    assert abs(c) < 5.0 / np.sqrt(zz.shape[0])
    return n


def test_mappo_policy_noise_stream_equals_restatement():
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    from oracle import philox_ref
    cfg = dict(ENV_CONFIG, episode_length=40)
    B, seed, rank = 4096, 11, 3                      # env_offset = 3 * 4096: global environment ids, not local ones
    tr = MAPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=B, rank=rank, world_size=1, verbose=False)
    m = tr.mappo
    rs = np.random.RandomState(0)
    log_std = rs.uniform(-0.7, 0.4, size=m.action_dim).astype(np.float32)
    with torch.no_grad():
        m.actor.log_std.copy_(torch.as_tensor(log_std))
    tr.train_episode(explore=True, record_buffer=True)           # noise=None: the in-kernel stream
    E, _ = m.buffer.totals()
    t = m.buffer.t
    step = _entry_steps(t["counters"], E)
    env = t["ent_env"][:E].cpu().numpy().astype(np.int64)
    agent = t["ent_agent"][:E].cpu().numpy().astype(np.int64)
    mu = m.actor(t["ent_obs"][:E]).cpu().numpy()
    eps = philox_ref.policy_noise_np(seed, tr.env_offset + env, tr.policy_episode, step, agent, m.action_dim, agent_shift=4)
    n = _check(t["ent_action"][:E].cpu().numpy(), t["ent_logp"][:E].cpu().numpy(), mu, log_std, eps, "MAPPO")
    assert n >= 1_000_000
    # the restated draws themselves are standard normal too (not only the kernel's)
    assert abs(eps.mean()) < 5.0 / np.sqrt(eps.size) and abs(eps.var() - 1.0) < 5.0 * np.sqrt(2.0 / eps.size)
    # a second episode uses a new episode index: fresh draws for the same (env, step, agent)
    first = t["ent_action"][:64].clone()
    m.buffer.clear()
    tr.train_episode(explore=True, record_buffer=True)
    assert not torch.equal(first, t["ent_action"][:64])


def test_grouped_policy_noise_stream_equals_restatement():
    """em_gmlp_policy (IPPO: one actor per decider): block index (agent << 8) | (component // 4)."""
    from marl_amd import gmlp
    from marl_amd.config import SCALED_ENV_CONFIG
    from marl_amd.ippo import IPPOTrainer
    from oracle import philox_ref
    cfg = dict(SCALED_ENV_CONFIG)
    cfg.update(n_nodes=8, n_deciders=24, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9, episode_length=30)
    B, seed = 1024, 5
    tr = IPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=B, rank=1, world_size=1, verbose=False)
    a = tr.algo
    rs = np.random.RandomState(1)
    log_std = rs.uniform(-0.5, 0.3, size=(a.G, a.S + 1)).astype(np.float32)
    o, n_ = a.actor.offsets["log_std"]
    a.actor.flat[:, o:o + n_] = torch.as_tensor(log_std).cuda()
    tr.train_episode(explore=True, record_buffer=True)
    E = a.buffer.total()
    t = a.buffer.t
    base = t["counters"].cpu().numpy()
    step = np.searchsorted(base, np.arange(E), side="right") - 1
    env = t["ent_env"][:E].cpu().numpy().astype(np.int64)
    agent = t["ent_agent"][:E].cpu().numpy().astype(np.int64)
    off, perm = gmlp.group_index(t["ent_agent"][:E], a.G)
    mu = gmlp.forward(a.actor, t["ent_obs"][:E].contiguous(), off, perm, 1).cpu().numpy()
    eps = philox_ref.policy_noise_np(seed, tr.env_offset + env, tr.policy_episode, step, agent, a.S + 1, agent_shift=8)
    action, logp = t["ent_action"][:E].cpu().numpy(), t["ent_logp"][:E].cpu().numpy()
    sigma = np.exp(log_std)[agent]
    want = mu + sigma * eps
    err = np.abs(action - want)
    print(f"IPPO: {E} entries; max |a - (mu + sigma eps)| = {err.max():.3e}")
    assert (err <= 5e-6 + 2e-6 * np.abs(want)).all()
    a64, m64, s64 = action.astype(np.float64), mu.astype(np.float64), sigma.astype(np.float64)
    lp = (-(a64 - m64) ** 2 / (2.0 * s64 ** 2) - np.log(s64) - HALF_LOG_2PI).sum(axis=1)
    np.testing.assert_allclose(logp, lp, rtol=1e-5, atol=1e-5)
    z = ((a64 - m64) / s64).ravel()
    assert abs(z.mean()) < 5.0 / np.sqrt(z.size) and abs(z.var() - 1.0) < 5.0 * np.sqrt(2.0 / z.size)
