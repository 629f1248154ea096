"""Container-only tool: known-answer vectors from a checkpoint the reference ships (mappo_seed_0.pt, written by
MAPPO.save, mappo_algorithm.py:347-355, after 20,000 Adam steps).  TEST INFRASTRUCTURE.

The checkpoint is read with torch.load(..., weights_only=True) (nothing in the file is executed), the IMPORTED reference
networks (MAPPOActor / MAPPOCritic, mappo_algorithm.py:16-90) are evaluated on fixed inputs, and two fixtures are written:

  tests/golden/ckpt_seed0.pt    the same checkpoint re-saved from plain tensors / floats (the reference's layout and keys)
  tests/golden/ckpt_seed0.npz   inputs (64 observations, 64 states taken from the recorded seed-0 episode, plus scaled
                                random rows) and the reference's outputs: mu, log_std, log-prob and entropy of fixed
                                actions, critic values, get_values (de-normalised), and the Adam bookkeeping

    python oracle/make_golden_ckpt.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.ref_loader import REFERENCE_ROOT, load_reference  # noqa: E402


def plain(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu().clone().contiguous()
    if isinstance(obj, dict):
        return {k: plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [plain(v) for v in obj]
    return obj


def main():
    _, algo, _ = load_reference()
    src = os.path.join(REFERENCE_ROOT, "mappo_seed_0.pt")
    ck = torch.load(src, map_location="cpu", weights_only=True)
    assert set(ck) == {"actor", "critic", "actor_optimizer", "critic_optimizer", "value_normalizer"}, sorted(ck)
    torch.manual_seed(0)
    m = algo.MAPPO(obs_dim=14, state_dim=140, action_dim=11, n_agents=10, device="cpu")
    m.actor.load_state_dict(ck["actor"])
    m.critic.load_state_dict(ck["critic"])
    m.actor_optimizer.load_state_dict(ck["actor_optimizer"])
    m.critic_optimizer.load_state_dict(ck["critic_optimizer"])
    m.value_normalizer.load_state_dict(ck["value_normalizer"])

    from tests.test_oracle_mappo import flatten_buffer, load
    ep_obs, ep_states, _ = flatten_buffer(load())          # the recorded seed-0 episode (tests/golden/mappo_seed0_T200.npz)
    rs = np.random.RandomState(2025)
    obs = np.concatenate([ep_obs[:48], rs.normal(0, 2.0, size=(16, 14)).astype(np.float32)]).astype(np.float32)
    states = np.concatenate([ep_states[:48], rs.normal(0, 2.0, size=(16, 140)).astype(np.float32)]).astype(np.float32)
    actions = rs.normal(0, 1.0, size=(64, 11)).astype(np.float32)
    with torch.no_grad():
        mu = m.actor(torch.as_tensor(obs))
        _, logp, ent = m.actor.get_action_log_prob(torch.as_tensor(obs), torch.as_tensor(actions))
        v = m.critic(torch.as_tensor(states)).squeeze(-1)
        get_values = np.array([m.get_values(s) for s in states[:8]], np.float64)
    st = ck["actor_optimizer"]["state"]
    out = {
        "obs": obs, "states": states, "actions": actions,
        "mu": mu.numpy(), "log_std": m.actor.log_std.detach().numpy().copy(),
        "logp": logp.numpy(), "entropy": ent.numpy(), "values": v.numpy(), "get_values": get_values,
        "adam_step": np.array(float(st[0]["step"])), "adam_lr": np.array(ck["actor_optimizer"]["param_groups"][0]["lr"]),
        # torch.optim.Adam numbers its state by module.parameters() order: the actor's own parameter log_std comes first
        # (index 0), fc1.weight is index 1; the critic has no parameter of its own, so its fc1.weight is index 0
        "actor_exp_avg_log_std": st[0]["exp_avg"].numpy(), "actor_exp_avg_fc1": st[1]["exp_avg"].numpy(), "critic_exp_avg_sq_fc1": ck["critic_optimizer"]["state"][0]["exp_avg_sq"].numpy(),
        "norm": np.array([float(ck["value_normalizer"]["running_mean"].reshape(-1)[0]),
                          float(ck["value_normalizer"]["running_var"].reshape(-1)[0]),
                          float(ck["value_normalizer"]["count"])]),
    }
    for k, t in ck["actor"].items():
        out["actor." + k] = t.numpy()
    for k, t in ck["critic"].items():
        out["critic." + k] = t.numpy()
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ckpt_seed0.npz"), **out)
    torch.save(plain(ck), os.path.join(ROOT, "tests", "golden", "ckpt_seed0.pt"))
    print("wrote tests/golden/ckpt_seed0.{npz,pt}: adam step", float(st[0]["step"]), "mu[0,:3]", mu[0, :3].tolist(),
          "v[:3]", v[:3].tolist())


if __name__ == "__main__":
    main()
