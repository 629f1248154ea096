"""Checkpoint interchange (mappo_algorithm.py:347-364): a checkpoint written here has the reference's keys and
tensor names and loads back; a state dict in torch.optim.Adam's layout round-trips through HipAdam."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_save_load_roundtrip_and_reference_layout(tmp_path):
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG)
    cfg["episode_length"] = 30
    tr = MAPPOTrainer(cfg, seed=1, device="cuda:0", n_envs=4, verbose=False)
    tr.train_episode()
    tr.mappo.update()
    path = str(tmp_path / "mappo_seed_1.pt")
    tr.save_model(path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert set(ck) == {"actor", "critic", "actor_optimizer", "critic_optimizer", "value_normalizer"}
    assert {k: tuple(v.shape) for k, v in ck["actor"].items()} == {
        "log_std": (11,), "fc1.weight": (64, 14), "fc1.bias": (64,), "fc2.weight": (64, 64), "fc2.bias": (64,),
        "mu.weight": (11, 64), "mu.bias": (11,)}
    assert {k: tuple(v.shape) for k, v in ck["critic"].items()} == {
        "fc1.weight": (64, 140), "fc1.bias": (64,), "fc2.weight": (64, 64), "fc2.bias": (64,),
        "value_out.weight": (1, 64), "value_out.bias": (1,)}
    # torch.optim.Adam accepts the optimizer state as is (same parameter order as module.parameters())
    ref_actor = torch.nn.ParameterList([torch.nn.Parameter(v.clone()) for v in ck["actor"].values()])
    opt = torch.optim.Adam(ref_actor.parameters(), lr=3e-4)
    opt.load_state_dict(ck["actor_optimizer"])
    assert float(opt.state_dict()["state"][0]["step"]) == 4.0
    tr2 = MAPPOTrainer(cfg, seed=9, device="cuda:0", n_envs=4, verbose=False)
    tr2.load_model(path)
    assert torch.equal(tr2.mappo.flat_params, tr.mappo.flat_params)
    assert torch.equal(tr2.mappo.actor_optimizer.exp_avg, tr.mappo.actor_optimizer.exp_avg)
    assert tr2.mappo.critic_optimizer.step_count == 4
    # both continue identically
    for t in (tr, tr2):
        t.env.reset(seed=5)
        t.policy_episode = 7
        t.seed = 1                       # policy-noise key
        t.env.global_step_counter = 100  # same exploration schedule position
        t.train_episode()
        t.mappo.update()
    assert torch.equal(tr2.mappo.flat_params, tr.mappo.flat_params)


def test_reference_checkpoint_known_answers():
    """A checkpoint the reference SHIPS (mappo_seed_0.pt after 20,000 Adam steps; re-saved from plain tensors by
    oracle/make_golden_ckpt.py as tests/golden/ckpt_seed0.pt) loads through MAPPO.load (mappo_algorithm.py:357-364), and the
    HIP kernels reproduce what the imported reference networks computed from it on 64 fixed inputs: mu, log-prob and entropy
    of fixed actions, critic values, get_values; Adam's step and moments are restored."""
    import os
    from marl_amd.mappo import MAPPO
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "ckpt_seed0.npz"), allow_pickle=False)
    m = MAPPO(obs_dim=14, state_dim=140, action_dim=11, n_agents=10, device="cuda:0", n_envs=1, horizon=4)
    m.load(os.path.join(here, "ckpt_seed0.pt"))
    for k in g.files:
        if k.startswith("actor.") or k.startswith("critic."):
            mod, _, name = k.partition(".")
            assert np.array_equal(dict(getattr(m, mod).named_parameters())[name].detach().cpu().numpy(), g[k]), k
    assert m.actor_optimizer.step_count == m.critic_optimizer.step_count == int(g["adam_step"]) == 20000
    assert m.actor_optimizer.param_groups[0]["lr"] == float(g["adam_lr"])
    off, n = m.actor._slices["fc1.weight"]
    assert np.array_equal(m.actor_optimizer.exp_avg[off:off + n].view(64, 14).cpu().numpy(), g["actor_exp_avg_fc1"])
    off, n = m.actor._slices["log_std"]
    assert np.array_equal(m.actor_optimizer.exp_avg[off:off + n].cpu().numpy(), g["actor_exp_avg_log_std"])
    off, n = m.critic._slices["fc1.weight"]
    assert np.array_equal(m.critic_optimizer.exp_avg_sq[off:off + n].view(64, 140).cpu().numpy(), g["critic_exp_avg_sq_fc1"])
    mean, var = m._norm()
    assert (mean, var) == (float(g["norm"][0]), float(g["norm"][1])) == (0.0, 1.0)
    obs, states, actions = (torch.as_tensor(g[k]).cuda() for k in ("obs", "states", "actions"))
    mu = m.actor(obs).cpu().numpy()
    v = m.critic(states).squeeze(-1).cpu().numpy()
    # trained weights give outputs of magnitude 10..200: 1e-5 relative with a small absolute floor for components near 0
    np.testing.assert_allclose(mu, g["mu"], rtol=1e-5, atol=1e-5 * np.abs(g["mu"]).max())
    np.testing.assert_allclose(v, g["values"], rtol=1e-5, atol=1e-5 * np.abs(g["values"]).max())
    assert np.array_equal(m.actor.log_std.detach().cpu().numpy(), g["log_std"])
    _, lp, ent = m.actor.get_action_log_prob(obs, actions)
    np.testing.assert_allclose(lp.cpu().numpy(), g["logp"], rtol=2e-5)        # sums of ~1e3-sized terms of both signs
    np.testing.assert_allclose(ent.cpu().numpy(), g["entropy"], rtol=1e-6)
    gv = np.array([m.get_values(g["states"][i]) for i in range(8)])
    np.testing.assert_allclose(gv, g["get_values"], rtol=1e-5)
    print("max rel err: mu %.2e  values %.2e  logp %.2e" % (
        np.max(np.abs(mu - g["mu"])) / np.abs(g["mu"]).max(), np.max(np.abs(v - g["values"]) / np.abs(g["values"])),
        np.max(np.abs(lp.cpu().numpy() - g["logp"]) / np.abs(g["logp"]))))
