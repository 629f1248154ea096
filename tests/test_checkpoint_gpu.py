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
