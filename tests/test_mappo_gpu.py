"""GPU parity of the MAPPO path (rollout, GAE, PPO update on the HIP kernels, through the C ABI) against the golden
episode + update recorded from the imported reference MAPPOTrainer(seed=0) (tests/golden/mappo_seed0_T200.npz).
Tolerance: 1e-5 relative in fp32 (north_star); environment decisions bit-exact."""
import numpy as np
import pytest
import torch

from tests.golden_util import assert_losses_close
from tests.test_env_gpu import TAPE_MAP
from tests.test_oracle_mappo import flatten_buffer, load

pytestmark = pytest.mark.gpu


def _trainer(g, n_envs=1):
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG)
    cfg["episode_length"] = int(g["episode_length"])
    return MAPPOTrainer(cfg, seed=int(g["seed"]), device="cuda:0", n_envs=n_envs, verbose=False)


def _tapes(g, device, B=1):
    from marl_amd.env import DeviceTape
    T = int(g["episode_length"])
    tape = {k[5:]: g[k] for k in g if k.startswith("tape_")}
    return [DeviceTape({TAPE_MAP[k]: np.stack([tape[k][s]] * B) for k in TAPE_MAP}, device) for s in range(T + 1)]


def _params(m):
    out = {f"actor.{k}": v.detach().cpu().numpy() for k, v in m.actor.state_dict().items()}
    out.update({f"critic.{k}": v.detach().cpu().numpy() for k, v in m.critic.state_dict().items()})
    return out


def _rollout(g, tr):
    tr.env.global_step_counter = int(g["explore_counter0"])
    noise = torch.as_tensor(g["noise"]).cuda().contiguous()
    return tr.train_episode(explore=True, record_buffer=True, tapes=_tapes(g, tr.device), noise=noise)


def test_seeded_init_equals_reference():
    g = load()
    tr = _trainer(g)
    for k, v in _params(tr.mappo).items():
        assert np.array_equal(v, g["init"][k]), k


def _apply_overrides(m, g):
    """The attribute overrides of run_mappo_batch.py:255-261, with the values the golden update was recorded with."""
    lr_a, lr_c, clip, ent, n_iter = g["hyper"].tolist()
    m.actor_optimizer.param_groups[0]["lr"] = lr_a
    m.critic_optimizer.param_groups[0]["lr"] = lr_c
    m.clip_param, m.entropy_coeff, m.num_sgd_iter = clip, ent, int(n_iter)
    m.mini_batch_size = 256          # inert in the reference too


@pytest.mark.parametrize("name", ["mappo_seed0_T200", "mappo_seed0_T1000"])
def test_rollout_gae_update_parity(name):
    """T200: a short episode; T1000: configs[0] at its real length (reward -1787.630, 883 transitions) with a
    third update whose hyper-parameters are all overridden (distinct learning rates, clip 0.15, entropy 0.005, 6 epochs)."""
    g = load(name)
    obs_flat, states_flat, step_of = flatten_buffer(g)
    L = len(step_of)
    cases = [(1, "p1", "loss1"), (4, "p4", "loss4")] + ([(None, "ph", "lossh")] if g["ph"] else [])
    for n_iter, pkey, lkey in cases:
        tr = _trainer(g)
        m = tr.mappo
        if n_iter is None:
            _apply_overrides(m, g)
        else:
            m.num_sgd_iter = n_iter
        stats = _rollout(g, tr)
        ro = m.buffer
        E, S = ro.totals()
        assert E == L and S == len(np.unique(step_of))
        t = {k: v.cpu().numpy() for k, v in ro.t.items() if v is not None}
        # environment side: bit-exact (observations prove that every placement/resource decision matched)
        assert np.array_equal(t["ent_obs"][:E], obs_flat)
        assert np.array_equal(ro.state_rows(S).cpu().numpy(), states_flat[np.unique(step_of, return_index=True)[1]])
        rew = np.where(g["step_emitted"], g["step_reward"], 0.0)
        np.testing.assert_allclose(t["reward_tb"][:, 0], rew, rtol=1e-6, atol=0)
        assert abs(stats["episode_reward"] - float(g["episode_reward"])) <= 1e-6 * abs(float(g["episode_reward"]))
        # policy side
        np.testing.assert_allclose(t["ent_action"][:E], g["buf_actions"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(t["ent_logp"][:E], g["buf_logp"], rtol=1e-5, atol=1e-5)
        vals = t["value_tb"][:, 0][step_of]
        np.testing.assert_allclose(vals, g["buf_values"], rtol=1e-5, atol=1e-5)
        assert np.array_equal(t["done_tb"][:, 0][step_of], g["buf_dones"])
        # GAE
        m.compute_gae()
        torch.cuda.synchronize()
        adv, ret = ro.t["ent_adv"][:E].cpu().numpy(), ro.t["ent_ret"][:E].cpu().numpy()
        np.testing.assert_allclose(adv, g["gae_adv"], rtol=1e-5, atol=0)      # north_star: advantages 1e-5 relative
        np.testing.assert_allclose(ret, g["gae_ret"], rtol=1e-5, atol=0)
        mom = m.moments.cpu().numpy()
        assert mom[0] == E
        np.testing.assert_allclose(mom[1] / E, np.float32(g["gae_adv"]).astype(np.float64).mean(), rtol=1e-5)
        # update
        out = m.update()
        assert_losses_close([out["actor_loss"], out["critic_loss"], out["entropy"]], g[lkey], f"golden update x{n_iter}")
        got = _params(m)
        for k, v in g[pkey].items():
            np.testing.assert_allclose(got[k], v, rtol=1e-4, atol=2e-6, err_msg=f"{pkey}:{k}")


def test_batched_rollout_replicas_match_single():
    """B identical tapes + identical noise per env => every environment reproduces the single-env episode."""
    g = load()
    B = 5
    tr = _trainer(g, n_envs=B)
    tr.env.global_step_counter = int(g["explore_counter0"])
    T = int(g["episode_length"])
    # entry order is (step, env, rank): replicate the noise rows accordingly
    obs_flat, _, step_of = flatten_buffer(g)
    noise = []
    for t in range(T):
        rows = g["noise"][step_of == t]
        noise.extend([rows] * B)
    noise = torch.as_tensor(np.concatenate([r for r in noise if len(r)])).cuda().contiguous()
    stats = tr.train_episode(explore=True, record_buffer=True, tapes=_tapes(g, tr.device, B), noise=noise)
    ro = tr.mappo.buffer
    E, S = ro.totals()
    assert E == B * len(step_of)
    rew = ro.t["reward_tb"].cpu().numpy()
    assert all(np.array_equal(rew[:, 0], rew[:, b]) for b in range(B))
    out = tr.mappo.update()
    # oracle: the torch-CPU restatement on the concatenation of the B replicated buffers (the unbiased std of the
    # advantage normalisation sees B*L samples, so this is not the single-episode golden)
    from oracle import mappo_oracle as mo
    obs_flat, states_flat, _ = flatten_buffer(g)
    rep = lambda a: np.concatenate([a] * B)  # noqa: E731
    batch = {"obs": rep(obs_flat), "states": rep(states_flat), "actions": rep(g["buf_actions"]),
             "old_logp": rep(g["buf_logp"]), "adv": rep(g["gae_adv"]), "returns": rep(g["gae_ret"])}
    want, losses, _, _, _ = mo.update(g["init"], batch, n_iter=4)
    assert_losses_close([out["actor_loss"], out["critic_loss"], out["entropy"]], np.mean(np.array(losses), axis=0),
                        "5 replicated environments")
    got = _params(tr.mappo)
    for k, v in want.items():
        np.testing.assert_allclose(got[k], v, rtol=2e-4, atol=4e-6, err_msg=k)


def test_update_interval_two_accumulates_episodes_like_the_list_buffer():
    """update_interval = 2 (mappo_trainer.py:281-282): two episodes per environment accumulate before update().
    The reference runs ONE GAE over the flattened list of both episodes (mappo_algorithm.py:284-287) with the done flag
    stored only on entries of a final step that has acting agents, so an episode whose last step is empty leaks into the
    next one; the device scan over the dense [2T][B] rows must give exactly what the pinned gae_flat restatement gives
    on each environment's flattened list."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    from oracle import mappo_oracle as mo
    cfg = dict(ENV_CONFIG, episode_length=70)
    B, T = 6, 70
    tr = MAPPOTrainer(cfg, seed=3, device="cuda:0", n_envs=B, verbose=False, update_interval=2)
    s1 = tr.train_episode(explore=True, record_buffer=True)
    tr.evaluate(num_episodes=B)                       # an evaluation in between must not disturb the recorded batch
    s2 = tr.train_episode(explore=True, record_buffer=True)
    with pytest.raises(RuntimeError):
        tr.train_episode(explore=True, record_buffer=True)     # storage full: the caller must update()
    m, ro = tr.mappo, tr.mappo.buffer
    E, S = ro.totals()
    assert ro.steps_recorded == 2 * T and E > 0
    m.compute_gae()
    torch.cuda.synchronize()
    t = {k: v.cpu().numpy() for k, v in ro.t.items() if v is not None}
    n_leaks = 0
    for b in range(B):
        rewards, values, dones, slots = [], [], [], []
        for row in range(2 * T):
            k, first = int(t["k_tb"][row, b]), int(t["first_tb"][row, b])
            for j in range(k):
                rewards.append(float(t["reward_tb"][row, b]))
                values.append(float(t["value_tb"][row, b]))
                dones.append(bool(t["done_tb"][row, b]))
                slots.append(first + j)
        if int(t["k_tb"][T - 1, b]) == 0:
            n_leaks += 1                                  # first episode ends without a terminal flag on any entry
        adv, ret = mo.gae_flat(rewards, values, dones)
        assert np.array_equal(t["ent_adv"][slots], adv.astype(np.float32)), b
        assert np.array_equal(t["ent_ret"][slots], ret.astype(np.float32)), b
    assert s1["episode_steps"] == T and np.isfinite(s2["episode_reward"])
    out = m.update()
    assert ro.steps_recorded == 0 and np.isfinite(out["critic_loss"])
    tr.train_episode(explore=True, record_buffer=True)     # storage is free again
    print("episodes that leak into the next one:", n_leaks, "of", B)


def test_states_by_reference_equal_compact_copies():
    """The two storage modes of the rollout (states copied compactly vs held by reference in the observation history, with
    the acting-agent totals counted by the environment kernel instead of a counting launch) record the same transitions and
    give the same update, bit for bit."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.mappo import RolloutStorage
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=50)
    B = 700                                      # three 256-environment chunks, the last one partial
    outs = []
    for by_ref in (True, False):
        tr = MAPPOTrainer(cfg, seed=4, device="cuda:0", n_envs=B, verbose=False)
        if not by_ref:
            tr.mappo.buffer = tr._train_storage = RolloutStorage(B, 10, 50, tr.device, by_reference=False)
        assert tr.mappo.buffer.by_reference == by_ref
        tr.train_episode(explore=True, record_buffer=True)
        ro = tr.mappo.buffer
        E, S = ro.totals()
        rec = {k: ro.t[k][:E].clone() for k in ("ent_obs", "ent_action", "ent_logp", "ent_env", "ent_agent")}
        rec.update({k: ro.t[k][:S].clone() for k in ("st_value", "st_env", "st_first", "st_k")})
        rec.update(states=ro.state_rows(S).clone(), k_tb=ro.t["k_tb"].clone(), first_tb=ro.t["first_tb"].clone(),
                   reward=ro.t["reward_tb"].clone(), counters=ro.t["counters"].clone())
        out = tr.mappo.update()
        rec["params"] = tr.mappo.flat_params.clone()
        outs.append((rec, out, E, S))
        ev = tr.evaluate(num_episodes=B)          # the one-step scratch storage is by reference too (two slices, ping-pong)
        assert np.isfinite(ev["avg_latency_ms"])
    (a, oa, Ea, Sa), (b, ob, Eb, Sb) = outs
    assert (Ea, Sa) == (Eb, Sb) and Ea > 1000 and oa == ob
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_two_wavefront_critic_backward_equals_the_one_wavefront_kernel(monkeypatch):
    """mlp3_bwd_kernel's SPLIT variant (the dW1 accumulators in a second wavefront per SIMD, hand-overs through LDS counters)
    takes every partial sum in the order of the one-wavefront kernel: same episode, same update, same parameters bit for bit,
    by reference (scattered state rows) and with compact copies, over several tiles per wavefront pair."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.mappo import RolloutStorage
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=120)
    B = 1500
    for by_ref in (True, False):
        outs = []
        for split in ("1", "0"):
            monkeypatch.setenv("EM_BWD_SPLIT", split)
            tr = MAPPOTrainer(cfg, seed=9, device="cuda:0", n_envs=B, verbose=False)
            if not by_ref:
                tr.mappo.buffer = tr._train_storage = RolloutStorage(B, 10, 120, tr.device, by_reference=False)
            for _ in range(2):                   # the second update starts from parameters the first one produced
                tr.train_episode(explore=True, record_buffer=True)
                out = tr.mappo.update()
            outs.append((tr.mappo.flat_params.clone(), out))
        (pa, oa), (pb, ob) = outs
        assert np.isfinite(oa["critic_loss"]) and oa == ob, (oa, ob)
        assert torch.equal(pa, pb)


@pytest.mark.parametrize("n_envs,explore,n_nodes", [(700, True, 10), (700, False, 10), (5, True, 10), (4099, True, 10),
                                                    (300, True, 6), (300, True, 8), (300, True, 2)])
def test_one_launch_index_and_policy_equals_the_two_launches(n_envs, explore, n_nodes):
    """em_rollout_policy_step (entries numbered, copied and the actor evaluated on them in one kernel) against
    em_rollout_index_totals followed by em_policy_step: every stored value, the actions handed to the environment and the
    update that follows are identical bit for bit (partial chunks, a batch smaller than one tile, evaluation mode)."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=40)
    if n_nodes != 10:       # the kernel's generic instantiation (node count read from the arguments)
        cfg.update(n_nodes=n_nodes, node_cpu_capacity=list(np.linspace(1.0e9, 3.0e9, n_nodes)))
    outs = []
    for fused in (True, False):
        tr = MAPPOTrainer(cfg, seed=9, device="cuda:0", n_envs=n_envs, verbose=False)
        tr.mappo.fuse_index_policy = fused
        tr.train_episode(explore=explore, record_buffer=True)
        ro = tr.mappo.buffer
        E, S = ro.totals()
        rec = {k: ro.t[k][:E].clone() for k in ("ent_obs", "ent_action", "ent_logp", "ent_env", "ent_agent")}
        rec.update({k: ro.t[k][:S].clone() for k in ("st_value", "st_env", "st_first", "st_k", "st_row")})
        rec.update(k_tb=ro.t["k_tb"].clone(), first_tb=ro.t["first_tb"].clone(), reward=ro.t["reward_tb"].clone(),
                   counters=ro.t["counters"].clone(), obs_tb=ro.t["obs_tb"].clone())
        out = tr.mappo.update()
        rec["params"] = tr.mappo.flat_params.clone()
        outs.append((rec, out, E, S))
    (a, oa, Ea, Sa), (b, ob, Eb, Sb) = outs
    assert (Ea, Sa) == (Eb, Sb) and Ea > 0 and oa == ob
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_one_launch_index_and_policy_rejects_what_it_cannot_do():
    from marl_amd import _lib
    from marl_amd.config import ENV_CONFIG
    from marl_amd.mappo import RolloutStorage
    from marl_amd.trainer import MAPPOTrainer
    import ctypes as C
    tr = MAPPOTrainer(dict(ENV_CONFIG, episode_length=5), seed=0, device="cuda:0", n_envs=8, verbose=False)
    m = tr.mappo
    L = _lib.lib()
    ro = m.buffer
    net = m.actor.net()
    z = torch.zeros(8, 10, dtype=torch.int32, device="cuda")
    zf = torch.zeros(8, 10, device="cuda")
    act = torch.zeros(8, dtype=torch.int64, device="cuda")
    rk = torch.full((8, 10), -1, dtype=torch.int8, device="cuda")
    tot = ro.chunk_totals[0]

    def call(storage, obs, totals):
        return L.em_rollout_policy_step(storage.ref, 0, act.data_ptr(), rk.data_ptr(), obs.data_ptr(),
                                        totals.data_ptr() if totals is not None else None, None, C.byref(net),
                                        m.actor.log_std.data_ptr(), None, 1, 0, 0, 0, 0, z.data_ptr(), zf.data_ptr(), None)
    assert call(ro, ro.obs_slice(0), tot) == 0
    assert call(ro, ro.obs_slice(0), None) != 0                                  # totals are required
    assert call(ro, torch.zeros(8, 10, 14, device="cuda"), tot) != 0             # not a slice of the history
    copies = RolloutStorage(8, 10, 5, tr.device, by_reference=False)
    assert call(copies, torch.zeros(8, 10, 14, device="cuda"), tot) != 0         # states copied: the two-call path
    assert b"by reference" in L.em_last_error()


@pytest.mark.parametrize("n_nodes", [5, 7, 9, 10])
def test_update_gradients_against_torch_autograd_for_other_node_counts(n_nodes):
    """One epoch of update() on a recorded batch (states by reference) against torch autograd in float64 on the same buffers,
    for critic input widths on every path of the backward kernel: 45 (three 16-wide blocks, register prefetch), 77 and 117
    (wide, rows not a multiple of four floats: neither staged nor prefetched - this path read the dW1 operand rows without
    the by-reference indirection until round 3), 140 (staged, two wavefronts per SIMD).  mappo_algorithm.py:292-339."""
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=40)
    if n_nodes != 10:
        cfg.update(n_nodes=n_nodes, node_cpu_capacity=list(np.linspace(1.0e9, 3.0e9, n_nodes)))
    tr = MAPPOTrainer(cfg, seed=11, device="cuda:0", n_envs=300, verbose=False)
    tr.train_episode(explore=True, record_buffer=True)
    m = tr.mappo
    m.num_sgd_iter = 1
    ro = m.buffer
    E, S = ro.totals()
    m.compute_gae()
    cnt, s1, s2 = m.moments.cpu().numpy()
    mean, std = s1 / cnt, np.sqrt((s2 - s1 * s1 / cnt) / (cnt - 1))
    f64 = lambda t: t.detach().clone().double()  # noqa: E731

    def mlp(x, p):
        h = torch.relu(x @ p[0].t() + p[1])
        h = torch.relu(h @ p[2].t() + p[3])
        return h @ p[4].t() + p[5]

    # actor: ratio == 1 in the first epoch, so the clipped surrogate's gradient is that of -mean(adv_n * ratio)
    a_names = [k for k, _ in m.actor.named_parameters()]
    a_par = {k: f64(p).requires_grad_(True) for k, p in m.actor.named_parameters()}
    adv = (f64(ro.t["ent_adv"][:E]) - mean) / (std + 1e-8)
    act, lp_old = f64(ro.t["ent_action"][:E]), f64(ro.t["ent_logp"][:E])
    ls = a_par["log_std"]
    mu = mlp(f64(ro.t["ent_obs"][:E]), [a_par[k] for k in a_names if k != "log_std"])
    lp = (-(act - mu) ** 2 / (2 * torch.exp(ls) ** 2) - ls - 0.5 * np.log(2 * np.pi)).sum(-1)
    ratio = torch.exp(lp - lp_old)
    loss_a = -torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).mean() - 0.01 * (0.5 + 0.5 * np.log(2 * np.pi) + ls).sum()
    loss_a.backward()
    # critic: value of every entry's state against the entry's return (ValueNormalizer is the identity up to 1e-8)
    c_names = [k for k, _ in m.critic.named_parameters()]
    c_par = {k: f64(p).requires_grad_(True) for k, p in m.critic.named_parameters()}
    v = mlp(f64(ro.state_rows(S)), [c_par[k] for k in c_names]).squeeze(1)
    first, k = ro.t["st_first"][:S].long(), ro.t["st_k"][:S].long()
    se = 0.0
    for j in range(int(k.max())):
        has = k > j
        se = se + ((v[has] - f64(ro.t["ent_ret"][(first + j)[has]]) / np.sqrt(1 + 1e-8)) ** 2).sum()
    (0.5 * 0.5 * se / E).backward()
    m.update()
    for module, par, base in ((m.actor, a_par, 0), (m.critic, c_par, m.n_actor)):
        for name, p in par.items():
            off, count = module._slices[name]
            got = m.flat_grads[base + off:base + off + count].double().view(p.shape)
            scale = float(p.grad.abs().max())
            assert float((got - p.grad).abs().max()) <= 2e-5 * max(scale, 1e-3), (name, scale)
