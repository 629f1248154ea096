"""Plain PyTorch-CPU float32 restatement of the grouped (per-agent) PPO arithmetic.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module.  PARITY UNPINNED against the reference: the reference's IPPO is RLlib's PPO
with one policy per agent (``MRRL.PY:611-638``; ``ray[rllib]`` 2.10.0 is not vendored and not installed here), and no
file of the reference pins its results.  The build therefore DEFINES independent PPO as "every agent runs the
reference's own MAPPO arithmetic on its own transitions" and this module restates exactly that from
``mappo_algorithm.py``: actor/critic MLPs (:16-90), Gaussian sample / log-prob / entropy (:62-71), action formatting
(``mappo_trainer.py:125-158``), GAE recursion (:241-262), advantage normalisation over the learner's batch (:293),
PPO-clip / value / entropy losses (:302-320), separate clip_grad_norm_ (:328-329) and Adam (:331-332).  With one group it
is the reference's shared-actor update, which ``oracle/mappo_oracle.py`` pins to the recorded reference run.
"""
from __future__ import annotations

import math

import numpy as np
import torch

HID = 64


def unpack(flat_row, in_dim, out_dim, with_log_std):
    """[w1|b1|w2|b2|w3|b3|(log_std)] -> dict of tensors (views)."""
    out, off = {}, 0
    for name, shape in (("w1", (HID, in_dim)), ("b1", (HID,)), ("w2", (HID, HID)), ("b2", (HID,)), ("w3", (out_dim, HID)),
                        ("b3", (out_dim,))) + ((("log_std", (out_dim,)),) if with_log_std else ()):
        n = int(np.prod(shape))
        out[name] = flat_row[off:off + n].view(shape)
        off += n
    return out


def mlp(p, x):
    h = torch.relu(x @ p["w1"].T + p["b1"])
    h = torch.relu(h @ p["w2"].T + p["b2"])
    return h @ p["w3"].T + p["b3"]


def forward(flat, in_dim, out_dim, with_log_std, x, agent):
    """y[row] = MLP_{agent[row]}(x[row])."""
    y = torch.zeros((x.shape[0], out_dim), dtype=torch.float32)
    for g in torch.unique(agent).tolist():
        idx = (agent == g).nonzero().flatten()
        y[idx] = mlp(unpack(flat[g], in_dim, out_dim, with_log_std), x[idx])
    return y


def sample(mu, log_std_rows, noise):
    """rsample + summed log-prob (mappo_algorithm.py:62-71)."""
    sd = torch.exp(log_std_rows)
    a = mu + noise * sd
    lp = (-((a - mu) ** 2) / (2 * sd * sd) - log_std_rows - 0.5 * math.log(2 * math.pi)).sum(-1)
    return a, lp


def format_actions(a, n_place):
    """mappo_trainer.py:125-158: placement = first argmax (0 if all ~0), resource share from tanh of the last component."""
    logits = a[:, :n_place]
    place = torch.where((logits.abs() <= 1e-8).all(dim=1), torch.zeros(a.shape[0], dtype=torch.int64), logits.argmax(dim=1))
    norm = ((torch.tanh(a[:, n_place].double()) + 1.0) / 2.0).clamp(0.0, 1.0)
    return place.to(torch.int32), (0.1 + 0.9 * norm).to(torch.float32)


def group_losses(p, x, action, old_logp, adv, clip, ecoef):
    """Actor loss of ONE learner on its own batch (mappo_algorithm.py:293,302-309,317-320)."""
    n = adv.numel()
    std = adv.std() if n > 1 else torch.zeros(())      # a learner with one transition: variance 0 (torch gives NaN)
    A = (adv - adv.mean()) / (std + 1e-8)
    mu = mlp(p, x)
    sd = torch.exp(p["log_std"])
    lp = (-((action - mu) ** 2) / (2 * sd * sd) - p["log_std"] - 0.5 * math.log(2 * math.pi)).sum(-1)
    ent = (0.5 + 0.5 * math.log(2 * math.pi) + p["log_std"]).sum().expand(n)
    ratio = torch.exp(lp - old_logp)
    s1, s2 = ratio * A, torch.clamp(ratio, 1 - clip, 1 + clip) * A
    actor_loss = -torch.min(s1, s2).mean()
    return actor_loss + ecoef * (-ent.mean()), torch.min(s1, s2).sum().item(), ent.sum().item()


def actor_grads(flat, in_dim, out_dim, x, action, old_logp, adv, agent, clip=0.2, ecoef=0.01):
    """Per-group gradient [G, n_per] and loss sums [G, 3] of the actor objective."""
    G = flat.shape[0]
    grads = torch.zeros_like(flat)
    loss3 = torch.zeros((G, 3))
    for g in range(G):
        idx = (agent == g).nonzero().flatten()
        if idx.numel() == 0:
            continue
        row = flat[g].clone().requires_grad_(True)
        loss, smin, sent = group_losses(unpack(row, in_dim, out_dim, True), x[idx], action[idx], old_logp[idx], adv[idx], clip, ecoef)
        loss.backward()
        grads[g] = row.grad
        loss3[g] = torch.tensor([smin, sent, float(idx.numel())])
    return grads, loss3


def critic_grads(flat, in_dim, x, ret, agent, vcoef=0.5, norm_mean=0.0, norm_var=1.0):
    """mappo_algorithm.py:303,312-314 per learner: vcoef * 0.5 * mean((norm(v) - norm(ret))^2)."""
    G = flat.shape[0]
    grads = torch.zeros_like(flat)
    loss3 = torch.zeros((G, 3))
    scale = math.sqrt(norm_var + 1e-8)
    for g in range(G):
        idx = (agent == g).nonzero().flatten()
        if idx.numel() == 0:
            continue
        row = flat[g].clone().requires_grad_(True)
        v = mlp(unpack(row, in_dim, 1, False), x[idx]).squeeze(-1)
        e = (v - norm_mean) / scale - (ret[idx] - norm_mean) / scale
        (vcoef * 0.5 * (e ** 2).mean()).backward()
        grads[g] = row.grad
        loss3[g] = torch.tensor([(e ** 2).sum().item(), 0.0, float(idx.numel())])
    return grads, loss3


def clip_adam(flat, grads, m, v, step, lr=3e-4, b1=0.9, b2=0.999, eps=1e-8, max_norm=0.5):
    """clip_grad_norm_ per network then torch.optim.Adam's single-tensor update (in place, returns the norms)."""
    norms = torch.zeros(flat.shape[0])
    for g in range(flat.shape[0]):
        gr = grads[g].clone()
        tn = gr.norm()
        norms[g] = tn
        gr = gr * torch.clamp(max_norm / (tn + 1e-6), max=1.0)
        m[g].lerp_(gr, 1 - b1)
        v[g].mul_(b2).addcmul_(gr, gr, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v[g].sqrt() / math.sqrt(bc2)).add_(eps)
        flat[g].addcdiv_(m[g], denom, value=-(lr / bc1))
    return norms


def gae_agents(reward_tb, done_tb, k_tb, first_tb, ent_agent, ent_value, gamma=0.99, lam=0.95):
    """One GAE chain per (environment, agent) over the agent's own transitions (the recursion of
    mappo_algorithm.py:250-258 in float64, next_value = 0 at the end, :284)."""
    T, B = k_tb.shape
    E = ent_agent.shape[0]
    adv = np.zeros(E, np.float32)
    ret = np.zeros(E, np.float32)
    for b in range(B):
        gae, vnext = {}, {}
        for t in range(T - 1, -1, -1):
            k = int(k_tb[t, b])
            if k == 0:
                continue
            r, nd = float(reward_tb[t, b]), 1.0 - float(done_tb[t, b])
            f = int(first_tb[t, b])
            for j in range(k):
                d = int(ent_agent[f + j])
                v = float(ent_value[f + j])
                delta = r + gamma * vnext.get(d, 0.0) * nd - v
                g = delta + gamma * lam * nd * gae.get(d, 0.0)
                gae[d], vnext[d] = g, v
                adv[f + j] = np.float32(g)
                ret[f + j] = np.float32(g + v)
    return adv, ret
