"""MAPPO on the scaled topology (marl_amd/mappo_scaled.py): shared actor + CENTRAL critic whose 1st layer is evaluated
sparsely (csrc/central_critic.hip), against a plain PyTorch-CPU float32 restatement that builds the DENSE concatenated
state of mappo_trainer.py:101-123 and runs the reference's update arithmetic (mappo_algorithm.py:241-332) on it.
Tolerances as tests/test_ippo_gpu.py."""
import math

import numpy as np
import pytest
import torch

from oracle import edge_env_oracle as eo
from oracle import ippo_oracle as io
from tests.test_ippo_gpu import _cfg, _params_close

pytestmark = pytest.mark.gpu


def _critic_parts(flat, D, W):
    n1 = 64 * D * W
    sizes = [("w1", (64, D * W)), ("b1", (64,)), ("w2", (64, 64)), ("b2", (64,)), ("w3", (1, 64)), ("b3", (1,))]
    out, off = {}, 0
    for name, shape in sizes:
        n = int(np.prod(shape))
        out[name] = flat[off:off + n].view(shape)
        off += n
    assert off == flat.numel() and n1 == out["w1"].numel()
    return out


def _dense_states(t, E, T, B, D, W):
    s = torch.zeros((T * B, D * W))
    for e in range(E):
        st, d = int(t["ent_state"][e]), int(t["ent_agent"][e])
        s[st, d * W:(d + 1) * W] = t["ent_obs"][e]
    return s


def _gae_flat(reward_tb, done_tb, k_tb, first_tb, value_tb, gamma=0.99, lam=0.95):
    """MAPPO.compute_gae (mappo_algorithm.py:241-262) over each environment's flattened per-agent list."""
    T, B = k_tb.shape
    E = int(k_tb.sum())
    adv, ret = np.zeros(E, np.float32), np.zeros(E, np.float32)
    for b in range(B):
        items = [(int(first_tb[t, b]) + j, t) for t in range(T) for j in range(int(k_tb[t, b]))]
        gae, nv = 0.0, 0.0
        for e, t in reversed(items):
            v, nd = float(value_tb[t, b]), 1.0 - float(done_tb[t, b])
            delta = float(reward_tb[t, b]) + gamma * nv * nd - v
            gae = delta + gamma * lam * nd * gae
            adv[e], ret[e] = np.float32(gae), np.float32(gae + v)
            nv = v
    return adv, ret


def test_scaled_mappo_episode_and_update_match_dense_restatement():
    from marl_amd.ippo import IPPOTrainer
    T, B, seed = 30, 5, 23
    cfg = _cfg(T)
    S, D = cfg["n_nodes"], cfg["n_deciders"]
    W = S + 4
    tr = IPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=B, algorithm="mappo", verbose=False)
    algo = tr.algo
    algo.num_sgd_iter = 2
    o, n = algo.actor.offsets["log_std"]
    algo.actor.flat[:, o:o + n] = -0.5
    fa0, fc0 = algo.actor.flat.cpu().clone(), algo.critic.flat.cpu().clone().view(-1)
    gen = torch.Generator().manual_seed(8)
    noise = torch.randn((algo.buffer.capacity_entries, S + 1), generator=gen)
    tr.train_episode(explore=True, record_buffer=True, noise=noise.to("cuda:0").contiguous())
    ro = algo.buffer
    E = ro.total()
    t = {k: v.cpu() for k, v in ro.t.items()}

    # environments replayed by the numpy restatement on the recorded actions
    envs = [eo.EdgeEnvOracle(cfg, eo.PhiloxDraws(seed, b, 1)) for b in range(B)]
    obs = [e.reset()[0] for e in envs]
    e_i = 0
    for step in range(T):
        for b, env in enumerate(envs):
            order = list(env.pending.keys())
            assert int(t["k_tb"][step, b]) == len(order)
            pl, rs = np.full(D, -1, np.int64), np.zeros(D, np.float32)
            for k, d in enumerate(order):
                assert int(t["ent_agent"][e_i]) == d and int(t["ent_state"][e_i]) == step * B + b
                assert np.array_equal(t["ent_obs"][e_i].numpy(), obs[b][d])
                p, r = io.format_actions(t["ent_action"][e_i:e_i + 1], S)
                pl[d], rs[d] = int(p[0]), float(r[0])
                e_i += 1
            out = env.step(pl, rs)
            obs[b] = out.obs
            assert float(t["reward_tb"][step, b]) == (out.reward if out.reward_emitted else 0.0)
    assert e_i == E

    # shared actor outputs and central-critic values of the recorded states
    x = t["ent_obs"][:E]
    zero = torch.zeros(E, dtype=torch.int64)
    mu = io.forward(fa0, W, S + 1, True, x, zero)
    a, lp = io.sample(mu, torch.full((E, S + 1), -0.5), noise[:E])
    np.testing.assert_allclose(t["ent_action"][:E].numpy(), a.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(t["ent_logp"][:E].numpy(), lp.numpy(), rtol=2e-5, atol=2e-4)
    states = _dense_states(t, E, T, B, D, W)
    cp = _critic_parts(fc0, D, W)
    v_dense = io.mlp(cp, states).squeeze(-1) * math.sqrt(1 + 1e-8)
    has = (t["k_tb"][:T].reshape(-1) > 0)
    np.testing.assert_allclose(t["value_tb"][:T].reshape(-1)[has].numpy(), v_dense[has].numpy(), rtol=1e-5, atol=1e-5)

    stats = algo.update()
    adv, ret = _gae_flat(t["reward_tb"].numpy(), t["done_tb"].numpy(), t["k_tb"].numpy(), t["first_tb"].numpy(), t["value_tb"].numpy())
    np.testing.assert_allclose(ro.t["ent_adv"][:E].cpu().numpy(), adv, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ro.t["ent_ret"][:E].cpu().numpy(), ret, rtol=1e-6, atol=1e-6)
    adv_t, ret_t = ro.t["ent_adv"][:E].cpu(), ro.t["ent_ret"][:E].cpu()

    # two epochs of the reference's update on the dense network
    fa, fc = fa0.clone(), fc0.clone().view(1, -1)
    ma, va, mc, vc = torch.zeros_like(fa), torch.zeros_like(fa), torch.zeros_like(fc), torch.zeros_like(fc)
    ent_state = t["ent_state"][:E].long()
    sums = np.zeros(2)
    for step in (1, 2):
        ga, la = io.actor_grads(fa, W, S + 1, x, t["ent_action"][:E], t["ent_logp"][:E], adv_t, zero)
        row = fc[0].clone().requires_grad_(True)
        v = io.mlp(_critic_parts(row, D, W), states).squeeze(-1)[ent_state]      # one evaluation per transition (:303)
        e = v / math.sqrt(1 + 1e-8) - ret_t / math.sqrt(1 + 1e-8)
        (0.5 * 0.5 * (e ** 2).mean()).backward()
        io.clip_adam(fa, ga, ma, va, step)
        io.clip_adam(fc, row.grad.view(1, -1), mc, vc, step)
        sums += [la[:, 0].sum().item(), (e ** 2).sum().item()]
    _params_close(algo.actor.flat.cpu().numpy(), fa.numpy())
    _params_close(algo.critic.flat.cpu().numpy(), fc.numpy())
    assert stats["transitions"] == E
    from tests.golden_util import assert_losses_close
    assert_losses_close([stats["actor_loss"], stats["critic_loss"], 1.0], [-sums[0] / (2 * E), 0.5 * sums[1] / (2 * E), 1.0],
                        "two epochs vs the torch restatement")


def test_scaled_mappo_64x256_smoke():
    from marl_amd.config import SCALED_ENV_CONFIG
    from marl_amd.ippo import IPPOTrainer
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 25
    tr = IPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=96, algorithm="mappo", verbose=False)
    before = tr.algo.critic.flat.clone()
    st = tr.train_episode()
    up = tr.algo.update()
    assert np.isfinite(st["episode_reward"]) and up["transitions"] > 96 * 25
    assert np.isfinite(up["actor_loss"]) and np.isfinite(up["critic_loss"])
    assert bool(torch.isfinite(tr.algo.critic.flat).all()) and not torch.equal(before, tr.algo.critic.flat)
    ev = tr.evaluate(num_episodes=96)
    assert all(np.isfinite(v) for v in ev.values())


@pytest.mark.parametrize("algorithm", ["mappo", "ippo"])
def test_deferred_critic_equals_the_per_step_critic(algorithm):
    """The critic of a recorded rollout evaluated once at the end of the episode (compute_values; the steps are then indexed
    as one group for the shared actor) gives the values, advantages and parameters of the per-step critic, bit for bit."""
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = 30
    outs = []
    for defer in (True, False):
        tr = IPPOTrainer(cfg, seed=11, device="cuda:0", n_envs=700, algorithm=algorithm, verbose=False)
        tr.algo.defer_values = defer
        tr.train_episode(explore=True, record_buffer=True)
        ro = tr.algo.buffer
        E = ro.total()
        rec = {k: ro.t[k][:E].clone() for k in ("ent_obs", "ent_action", "ent_logp", "ent_agent", "ent_env")}
        key = "value_tb" if algorithm == "mappo" else "ent_value"
        vals = ro.t[key].clone() if algorithm == "mappo" else ro.t[key][:E].clone()
        k_tb = ro.t["k_tb"].clone()
        up = tr.algo.update()
        outs.append((E, rec, vals, k_tb, up, tr.algo.actor.flat.clone(), tr.algo.critic.flat.clone()))
    (Ea, ra, va, ka, ua, aa, ca), (Eb, rb, vb, kb, ub, ab, cb) = outs
    assert Ea == Eb and Ea > 1000 and torch.equal(ka, kb)
    for k in ra:
        assert torch.equal(ra[k], rb[k]), k
    if algorithm == "mappo":
        has = ka > 0                             # states without an entry keep whatever the storage held
        assert torch.equal(va[:ka.shape[0]][has], vb[:kb.shape[0]][has])
    else:
        assert torch.equal(va, vb)
    assert ua == ub and torch.equal(aa, ab) and torch.equal(ca, cb)
