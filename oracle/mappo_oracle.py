"""Plain PyTorch-CPU float32 restatement of the MAPPO arithmetic.  TEST INFRASTRUCTURE ONLY.

Restates (functionally, on explicit parameter dictionaries) what the reference computes in
``mappo_algorithm.py``: actor/critic MLPs (:16-90), diagonal-Gaussian sampling and log-prob (:62-71),
the sequential float64 GAE (:241-262) and one ``update()`` (:264-345: global advantage normalisation,
PPO-clip, value MSE through the inert normaliser :380-386, entropy bonus, separate grad-clip, Adam),
plus the trainer's action formatting (``mappo_trainer.py:125-158``).

Pinned against ``tests/golden/mappo_seed0_T200.npz`` (made from the imported reference by
``oracle/make_golden.py``) in ``tests/test_oracle_mappo.py``.
"""
from __future__ import annotations

import math

import numpy as np
import torch

ACTOR_KEYS = ("actor.fc1.weight", "actor.fc1.bias", "actor.fc2.weight", "actor.fc2.bias", "actor.mu.weight",
              "actor.mu.bias", "actor.log_std")
CRITIC_KEYS = ("critic.fc1.weight", "critic.fc1.bias", "critic.fc2.weight", "critic.fc2.bias",
               "critic.value_out.weight", "critic.value_out.bias")
NORM_SCALE = math.sqrt(1.0 + 1e-8)   # ValueNormalizer with mean 0 / var 1 forever (update() is never called)


def to_torch(params: dict) -> dict:
    return {k: torch.as_tensor(np.asarray(v), dtype=torch.float32).clone() for k, v in params.items()}


def actor_mu(p: dict, obs: torch.Tensor) -> torch.Tensor:
    h = torch.relu(obs @ p["actor.fc1.weight"].T + p["actor.fc1.bias"])
    h = torch.relu(h @ p["actor.fc2.weight"].T + p["actor.fc2.bias"])
    return h @ p["actor.mu.weight"].T + p["actor.mu.bias"]


def critic_v(p: dict, state: torch.Tensor) -> torch.Tensor:
    h = torch.relu(state @ p["critic.fc1.weight"].T + p["critic.fc1.bias"])
    h = torch.relu(h @ p["critic.fc2.weight"].T + p["critic.fc2.bias"])
    return (h @ p["critic.value_out.weight"].T + p["critic.value_out.bias"]).squeeze(-1)


def gaussian_logp_entropy(mu, log_std, action):
    std = torch.exp(log_std)
    logp = (-((action - mu) ** 2) / (2 * std * std) - log_std - 0.5 * math.log(2 * math.pi)).sum(-1)
    ent = (0.5 + 0.5 * math.log(2 * math.pi) + log_std).sum(-1).expand(mu.shape[0])
    return logp, ent


def sample(p: dict, obs: torch.Tensor, noise: torch.Tensor):
    """a = mu + sigma*eps (rsample), logp as the reference stores it."""
    mu = actor_mu(p, obs)
    a = mu + noise * torch.exp(p["actor.log_std"])
    logp, _ = gaussian_logp_entropy(mu, p["actor.log_std"], a)
    return a, logp, mu


def rollout_value(p: dict, state: torch.Tensor) -> torch.Tensor:
    """get_values: critic output de-normalised (mappo_algorithm.py:226-239)."""
    return critic_v(p, state) * torch.sqrt(torch.tensor(1.0 + 1e-8)) + 0.0


def format_env_action(a: np.ndarray, n: int):
    """mappo_trainer.py:125-158 for one action vector."""
    vec = np.asarray(a, np.float32).reshape(-1)
    logits = vec[:n]
    placement = 0 if np.allclose(logits, 0.0) else int(np.argmax(logits) % n)
    norm = (np.tanh(float(vec[-1])) + 1.0) / 2.0
    share = 0.1 + 0.9 * np.clip(norm, 0.0, 1.0)
    return placement, np.array([share], dtype=np.float32)[0]


def gae_flat(rewards, values, dones, gamma=0.99, lam=0.95, next_value=0.0):
    n = len(rewards)
    adv = [0.0] * n
    g = 0.0
    for t in reversed(range(n)):
        nv = next_value if t == n - 1 else values[t + 1]
        nd = 1 - int(dones[t])
        delta = rewards[t] + gamma * nv * nd - values[t]
        g = delta + gamma * lam * nd * g
        adv[t] = g
    ret = [a + v for a, v in zip(adv, values)]
    return np.array(adv), np.array(ret)


def normalise_advantages(adv32: torch.Tensor) -> torch.Tensor:
    return (adv32 - adv32.mean()) / (adv32.std() + 1e-8)


def ppo_losses(p, obs, states, actions, old_logp, adv_n, returns, clip=0.2):
    mu = actor_mu(p, obs)
    logp, ent = gaussian_logp_entropy(mu, p["actor.log_std"], actions)
    ratio = torch.exp(logp - old_logp)
    s1 = ratio * adv_n
    s2 = torch.clamp(ratio, 1 - clip, 1 + clip) * adv_n
    actor_loss = -torch.min(s1, s2).mean()
    v = critic_v(p, states)
    critic_loss = 0.5 * (((v - 0.0) / NORM_SCALE - (returns - 0.0) / NORM_SCALE) ** 2).mean()
    return actor_loss, critic_loss, ent.mean()


def update(params: dict, batch: dict, n_iter=4, lr_actor=3e-4, lr_critic=3e-4, clip=0.2, vcoef=0.5, ecoef=0.01,
           max_norm=0.5, adam_state=None):
    """One MAPPO.update() on explicit tensors.  batch: obs[L,14], states[L,140], actions[L,11], old_logp[L],
    adv[L] (un-normalised, float64 from GAE), returns[L].  Returns (new params, per-iteration losses, grads of it 0)."""
    p = {k: v.clone().requires_grad_(True) for k, v in to_torch(params).items()}
    a_par = [p[k] for k in ACTOR_KEYS]
    c_par = [p[k] for k in CRITIC_KEYS]
    opt_a = torch.optim.Adam(a_par, lr=lr_actor)
    opt_c = torch.optim.Adam(c_par, lr=lr_critic)
    if adam_state is not None:
        opt_a.load_state_dict(adam_state[0])
        opt_c.load_state_dict(adam_state[1])
    obs = torch.as_tensor(batch["obs"], dtype=torch.float32)
    states = torch.as_tensor(batch["states"], dtype=torch.float32)
    actions = torch.as_tensor(batch["actions"], dtype=torch.float32)
    old_logp = torch.as_tensor(np.asarray(batch["old_logp"]), dtype=torch.float32)
    adv = torch.as_tensor(np.asarray(batch["adv"]), dtype=torch.float32)
    returns = torch.as_tensor(np.asarray(batch["returns"]), dtype=torch.float32)
    adv_n = normalise_advantages(adv)
    losses, grads0 = [], None
    for it in range(n_iter):
        la, lc, ent = ppo_losses(p, obs, states, actions, old_logp, adv_n, returns, clip)
        loss = la + vcoef * lc + ecoef * (-ent)
        opt_a.zero_grad()
        opt_c.zero_grad()
        loss.backward()
        if it == 0:
            grads0 = {k: p[k].grad.detach().clone().numpy() for k in p}
        torch.nn.utils.clip_grad_norm_(a_par, max_norm)
        torch.nn.utils.clip_grad_norm_(c_par, max_norm)
        opt_a.step()
        opt_c.step()
        losses.append((la.item(), lc.item(), ent.item()))
    out = {k: v.detach().numpy().copy() for k, v in p.items()}
    return out, losses, grads0, adv_n.numpy(), (opt_a.state_dict(), opt_c.state_dict())
