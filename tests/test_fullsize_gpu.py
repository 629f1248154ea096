"""BASELINE.json configs[2] at FULL size: 65,536 environments x 1000 steps, recorded rollout + GAE + update on one GPU
(54 M transitions, 33 M state rows x 140 floats = 4.6 G elements: every buffer index past 2^32 elements is exercised).
What can be checked at this size without a CPU oracle that would run for hours:

  * bookkeeping: entries == sum of the per-(step, env) acting-agent counts, no overflow flag, counters consistent;
  * sampled environments (first, middle, last of the batch) are IDENTICAL, bit for bit, to a 4-environment run with the
    same global environment ids (the Philox streams of the environment and of the policy noise are keyed by the global
    id): observations, raw actions, log-probs, rewards, values, state rows, advantages and returns.  The small run is the
    one the golden tapes and the C restatement pin (tests/test_env_gpu.py, tests/test_mappo_gpu.py);
  * the update's loss sums over all 54 M transitions against a plain PyTorch fp32 evaluation of the same networks on the
    same buffers (chunked, on the GPU as a checker only), to north_star's 1e-5.
"""
import numpy as np
import pytest
import torch

from tests.golden_util import assert_losses_close

pytestmark = pytest.mark.gpu

B, T = 65536, 1000


def _trainer(n_envs, rank):
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    return MAPPOTrainer(dict(ENV_CONFIG), seed=0, device="cuda:0", n_envs=n_envs, rank=rank, world_size=1, verbose=False)


def _state_rows(t, idx):
    """State rows by index: the storage holds them by reference (rows of the observation history)."""
    flat = t["obs_tb"].view(-1, 140)
    return flat[t["st_row"][idx].long()]


def _env_rows(t, E, S, env):
    """Everything the storage holds about one environment, in recording order."""
    sel = (t["ent_env"][:E] == env).nonzero().squeeze(1)
    ssel = (t["st_env"][:S] == env).nonzero().squeeze(1)
    return {"obs": t["ent_obs"][sel], "action": t["ent_action"][sel], "logp": t["ent_logp"][sel], "agent": t["ent_agent"][sel],
            "adv": t["ent_adv"][sel], "ret": t["ent_ret"][sel], "state": _state_rows(t, ssel), "value": t["st_value"][ssel],
            "k": t["st_k"][ssel], "reward": t["reward_tb"][:, env], "done": t["done_tb"][:, env], "k_tb": t["k_tb"][:, env]}


def test_configs2_full_size_rollout_gae_update():
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2**30:
        pytest.skip("needs ~75 GB of free HBM (the MI355X has 288 GB)")
    tr = _trainer(B, 0)
    m = tr.mappo
    m.num_sgd_iter = 1
    stats = tr.train_episode(explore=True, record_buffer=True)
    assert not stats["storage_overflow"] and np.isfinite(stats["episode_reward"])
    ro = m.buffer
    E, S = ro.totals()
    t = ro.t
    # ---- bookkeeping --------------------------------------------------------------------------------------------------
    assert int(t["err_flag"].item()) == 0
    k_sum = int(t["k_tb"].sum(dtype=torch.int64).item())
    assert E == k_sum and S == int((t["k_tb"] > 0).sum().item())
    assert 40_000_000 < E < ro.capacity_entries and S * 140 > 2**32          # indexing past 2^32 elements is on the path
    c = t["counters"][:T + 1].cpu().numpy()
    assert np.array_equal(c[1:, 0] - c[:-1, 0], c[:-1, 1]) and np.array_equal(c[1:, 2] - c[:-1, 2], c[:-1, 3])
    assert np.array_equal(c[:-1, 1], t["k_tb"].sum(dim=1, dtype=torch.int64).cpu().numpy())
    assert int(t["st_k"][:S].sum(dtype=torch.int64).item()) == E
    assert bool((t["done_tb"][T - 1] == 1).all()) and int(t["done_tb"][:T - 1].sum().item()) == 0
    # the very last entry and state row are real rows (not whatever the allocation held)
    last_env = int(t["ent_env"][E - 1].item())
    assert 0 <= last_env < B and int(t["st_env"][S - 1].item()) == last_env
    assert int(t["st_first"][S - 1].item()) + int(t["st_k"][S - 1].item()) == E
    w = 14
    ag = int(t["ent_agent"][E - 1].item())
    assert torch.equal(_state_rows(t, torch.tensor([S - 1], device="cuda"))[0].view(10, w)[ag], t["ent_obs"][E - 1])
    assert int(t["st_row"][S - 1].item()) == (T - 1) * B + last_env          # row of (last step, env) in the history
    m.compute_gae()
    mom = m.moments.cpu().numpy()
    assert mom[0] == E
    # ---- sampled environments == small runs on the same global ids --------------------------------------------------------
    full = {}
    for rank in (0, 8191, 16383):                      # environments 0..3, 32764..32767, 65532..65535
        for j in (0, 3):
            env = 4 * rank + j
            full[env] = {k: v.clone() for k, v in _env_rows(t, E, S, env).items()}
    # ---- update at full size: loss sums against plain torch fp32 on the same buffers ---------------------------------------
    n, s1, s2 = mom
    mean = np.float32(s1 / n)
    std = np.float32(np.sqrt((s2 - s1 * s1 / n) / (n - 1)))
    with torch.no_grad():
        crit = [p.detach().clone() for p in (m.critic.fc1.weight, m.critic.fc1.bias, m.critic.fc2.weight, m.critic.fc2.bias,
                                               m.critic.value_out.weight, m.critic.value_out.bias)]
        log_std = m.actor.log_std.detach().clone()
        se = torch.zeros((), dtype=torch.float64, device="cuda")
        chunk = 1 << 21
        for lo in range(0, S, chunk):
            hi = min(S, lo + chunk)
            h = torch.relu(torch.addmm(crit[1], _state_rows(t, slice(lo, hi)), crit[0].t()))
            h = torch.relu(torch.addmm(crit[3], h, crit[2].t()))
            v = torch.addmm(crit[5], h, crit[4].t()).squeeze(1)
            first, k = t["st_first"][lo:hi].long(), t["st_k"][lo:hi].long()
            for j in range(int(k.max().item())):
                has = k > j
                r = t["ent_ret"][(first + j)[has]]
                se += (((v[has] - r) / np.float32(np.sqrt(1 + 1e-8))) ** 2).sum(dtype=torch.float64)
        adv_n = (t["ent_adv"][:E] - mean) / (std + np.float32(1e-8))
        want_actor = -float(adv_n.sum(dtype=torch.float64).item()) / E           # ratio == 1 in the first epoch
        want_entropy = float((0.5 + 0.5 * np.log(2 * np.pi) + log_std.double()).sum().item())
    out = m.update()
    assert m.last_update["transitions"] == E
    assert_losses_close([out["actor_loss"], out["critic_loss"], out["entropy"]],
                        [want_actor, 0.5 * float(se.item()) / E, want_entropy], "65,536 envs x 1000 steps, first epoch")
    assert bool(torch.isfinite(m.flat_params).all())
    del tr, m, ro, t
    torch.cuda.empty_cache()
    # ---- the small runs -------------------------------------------------------------------------------------------------------
    for rank in (0, 8191, 16383):
        small = _trainer(4, rank)
        small.env.reset(seed=0)        # the constructor's probe re-keys the stream to seed + attempt when 4 envs start empty
        small.train_episode(explore=True, record_buffer=True)
        sm = small.mappo
        e, s = sm.buffer.totals()
        sm.compute_gae()
        torch.cuda.synchronize()
        for j in (0, 3):
            got, want = full[4 * rank + j], _env_rows(sm.buffer.t, e, s, j)
            for key in want:
                assert torch.equal(got[key], want[key]), f"environment {4 * rank + j}: {key} differs from the 4-environment run"
            assert want["obs"].shape[0] > 500
        small.env.close()


@pytest.mark.parametrize("algorithm", ["ippo", "mappo"])
def test_configs4_full_size_rollout_gae_update(algorithm):
    """BASELINE.json configs[4] at FULL size (the bench's ippo_scaled / mappo_scaled legs): 64 servers x 256 deciders, 16,384
    environments x 1000 steps, 36.7 M transitions, deferred critic, two-wavefront backward kernels.  Checked here:
    bookkeeping (entries == sum of the per-(step, env) pending counts == the step counters, no overflow flag), the
    advantages and returns of sampled environments against the Python restatement of the GAE recursion on the recorded
    rewards / values (bit for bit), ret - adv == value, and an update that leaves finite, changed parameters and reports
    every transition."""
    from marl_amd.config import SCALED_ENV_CONFIG
    from marl_amd.ippo import IPPOTrainer
    from oracle import ippo_oracle as io
    from oracle import mappo_oracle as mo
    Bs, Ts = 16384, 1000
    tr = IPPOTrainer(dict(SCALED_ENV_CONFIG), seed=0, device="cuda:0", n_envs=Bs, algorithm=algorithm, verbose=False)
    st = tr.train_episode(explore=True, record_buffer=True)
    ro = tr.algo.buffer
    t = ro.t
    E = ro.total()
    assert st["episode_steps"] == Ts and int(t["err_flag"].item()) == 0
    assert E == int(t["k_tb"][:Ts].sum(dtype=torch.int64).item()) == int(t["counters"][Ts].item()) and E > 30_000_000
    before_a, before_c = tr.algo.actor.flat.clone(), tr.algo.critic.flat.clone()
    sample = (0, Bs // 2 - 1, Bs - 1)
    k_of = {b: t["k_tb"][:Ts, b].cpu().numpy() for b in sample}          # update() clears the step counts
    up = tr.algo.update()
    assert up["transitions"] == E and all(np.isfinite(up[k]) for k in ("actor_loss", "critic_loss", "entropy"))
    for p, q in ((tr.algo.actor.flat, before_a), (tr.algo.critic.flat, before_c)):
        assert bool(torch.isfinite(p).all()) and not torch.equal(p, q)
    for b in sample:
        k = k_of[b]
        first = t["first_tb"][:Ts, b].cpu().numpy().astype(np.int64)
        idx = np.concatenate([np.arange(f, f + kk) for f, kk in zip(first, k)]) if k.sum() else np.zeros(0, np.int64)
        sel = torch.as_tensor(idx, device=t["ent_adv"].device)
        adv, ret = t["ent_adv"][sel].cpu().numpy(), t["ent_ret"][sel].cpu().numpy()
        rew, done = t["reward_tb"][:Ts, b].cpu().numpy(), t["done_tb"][:Ts, b].cpu().numpy()
        local_first = np.concatenate([[0], np.cumsum(k)[:-1]]).astype(np.int64)
        assert len(idx) > 1000
        if algorithm == "ippo":
            val = t["ent_value"][sel].cpu().numpy()
            agent = t["ent_agent"][sel].cpu().numpy()
            want_adv, want_ret = io.gae_agents(rew[:, None], done[:, None], k[:, None], local_first[:, None], agent, val)
        else:
            # MAPPO.compute_gae over the environment's flattened list: every entry of a step carries the state's value
            v_state = t["value_tb"][:Ts, b].cpu().numpy()
            val = np.repeat(v_state, k)
            r_flat, d_flat = np.repeat(rew, k), np.repeat(done, k)
            a64, r64 = mo.gae_flat(list(r_flat), list(val.astype(np.float64)), list(d_flat))
            want_adv, want_ret = a64.astype(np.float32), r64.astype(np.float32)
        assert np.array_equal(adv, want_adv), (algorithm, b)
        assert np.array_equal(ret, want_ret), (algorithm, b)
