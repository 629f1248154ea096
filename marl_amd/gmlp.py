"""Grouped 3-layer MLPs (``csrc/gmlp.hip``): G networks of one shape in ONE flat float32 buffer ``[G, n_per]`` with the
per-network layout ``[w1 | b1 | w2 | b2 | w3 | b3 | (log_std)]`` (torch.nn.Linear layout inside), so per-agent Adam and
the data-parallel all-reduce work on one tensor.  Thin wrappers over the C ABI; no torch op is on the compute path."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib

HID = 64


def _stream(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class GroupedMLP:
    """``n_groups`` independent ``in_dim -> 64 -> 64 -> out_dim`` ReLU MLPs (MAPPOActor / MAPPOCritic of
    mappo_algorithm.py:16-90, one per group), initialised like ``torch.nn.Linear`` (U(-1/sqrt(fan_in), 1/sqrt(fan_in)))."""

    def __init__(self, n_groups, in_dim, out_dim, with_log_std, device, seed=0):
        self.G, self.in_dim, self.out_dim, self.with_log_std = int(n_groups), int(in_dim), int(out_dim), bool(with_log_std)
        self.device = torch.device(device)
        sizes = [("w1", HID * in_dim), ("b1", HID), ("w2", HID * HID), ("b2", HID), ("w3", out_dim * HID), ("b3", out_dim)]
        if with_log_std:
            sizes.append(("log_std", out_dim))
        self.offsets, off = {}, 0
        for name, n in sizes:
            self.offsets[name] = (off, n)
            off += n
        self.n_per = off
        gen = torch.Generator().manual_seed(int(seed))
        flat = torch.zeros((self.G, self.n_per), dtype=torch.float32)
        for name, fan_in in (("w1", in_dim), ("b1", in_dim), ("w2", HID), ("b2", HID), ("w3", HID), ("b3", HID)):
            o, n = self.offsets[name]
            bound = 1.0 / math.sqrt(fan_in)
            flat[:, o:o + n] = (torch.rand((self.G, n), generator=gen) * 2 - 1) * bound
        self.flat = flat.to(self.device).contiguous()

    def view(self, name):
        o, n = self.offsets[name]
        shape = {"w1": (HID, self.in_dim), "w2": (HID, HID), "w3": (self.out_dim, HID)}.get(name, (n,))
        return self.flat[:, o:o + n].view(self.G, *shape)

    def struct(self) -> _lib.EmGmlp:
        base = self.flat.data_ptr()
        p = {k: base + 4 * o for k, (o, _) in self.offsets.items()}
        return _lib.EmGmlp(p["w1"], p["b1"], p["w2"], p["b2"], p["w3"], p["b3"], self.n_per if self.G > 1 else 0,
                           self.in_dim, HID, self.out_dim, self.G)

    def log_std_ptr(self):
        return self.flat.data_ptr() + 4 * self.offsets["log_std"][0]

    def slab_floats(self) -> int:
        return int(_lib.lib().em_gmlp_slab_floats(self.in_dim, self.out_dim, int(self.with_log_std)))


def group_index(agent: torch.Tensor, n_groups: int):
    """Stable grouping of rows by network: (group_off int64 [G+1], perm int32 [rows]); rows of a group keep their order, so
    summation orders (and hence results) are reproducible.  ``em_group_index``: a counting sort on the device."""
    if agent.device.type != "cuda":
        raise RuntimeError("group_index runs on an MI355X (device='cuda:N'); there is no CPU path")
    key = agent if agent.dtype == torch.int32 and agent.is_contiguous() else agent.to(torch.int32).contiguous()
    rows = int(key.numel())
    L = _lib.lib()
    off = torch.empty(n_groups + 1, dtype=torch.int64, device=agent.device)
    perm = torch.empty(rows, dtype=torch.int32, device=agent.device)
    scratch = torch.empty(max(1, int(L.em_group_index_scratch_ints(rows, n_groups))), dtype=torch.int32, device=agent.device)
    with torch.cuda.device(agent.device):
        _lib.check(L.em_group_index(key.data_ptr() if rows else None, rows, int(n_groups), off.data_ptr(),
                                    perm.data_ptr() if rows else None, scratch.data_ptr(), _stream(agent.device)), "em_group_index")
    return off, perm


def _ptr(t):
    return t.data_ptr() if t is not None else None


def forward(net: GroupedMLP, x, group_off=None, perm=None, split=1):
    rows = x.shape[0]
    y = torch.empty((rows, net.out_dim), dtype=torch.float32, device=x.device)
    st = net.struct()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_gmlp_forward(C.byref(st), x.data_ptr(), x.shape[1], rows, _ptr(group_off), _ptr(perm),
                                              int(split), y.data_ptr(), _stream(x.device)), "em_gmlp_forward")
    return y


def policy(net: GroupedMLP, x, ent_env, ent_agent, ent_dst, placement, resource, ent_action, ent_logp, group_off=None,
           perm=None, split=1, noise=None, explore=True, seed=0, episode=0, rng_step=0, env_offset=0, x_src=None):
    """x_src (the environment's pending observation rows, [n_envs][EMS_PENDING_CAP][width]): the rows are read there, at
    ent_dst, and copied into x on the way (em_gmlp_policy_gather: the step's ems_rollout_index call passed no rows)."""
    st = net.struct()
    if x_src is not None:
        assert x_src.is_contiguous() and x_src.shape[-1] == x.shape[1] and x_src.dtype == x.dtype
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().em_gmlp_policy_gather(C.byref(st), net.log_std_ptr(), x_src.data_ptr(), x.data_ptr(), x.shape[1],
                                                        x.shape[0], _ptr(group_off), _ptr(perm), int(split), _ptr(noise),
                                                        int(bool(explore)), int(seed) & (2**64 - 1), int(episode), int(rng_step),
                                                        int(env_offset), ent_env.data_ptr(), ent_agent.data_ptr(),
                                                        ent_dst.data_ptr(), net.out_dim - 1, ent_action.data_ptr(),
                                                        ent_logp.data_ptr(), placement.data_ptr(), resource.data_ptr(),
                                                        _stream(x.device)), "em_gmlp_policy_gather")
        return
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_gmlp_policy(C.byref(st), net.log_std_ptr(), x.data_ptr(), x.shape[1], x.shape[0],
                                             _ptr(group_off), _ptr(perm), int(split), _ptr(noise), int(bool(explore)),
                                             int(seed) & (2**64 - 1), int(episode), int(rng_step), int(env_offset),
                                             ent_env.data_ptr(), ent_agent.data_ptr(), ent_dst.data_ptr(), net.out_dim - 1,
                                             ent_action.data_ptr(), ent_logp.data_ptr(), placement.data_ptr(),
                                             resource.data_ptr(), _stream(x.device)), "em_gmlp_policy")


def value(net: GroupedMLP, x, out, group_off=None, perm=None, split=1, norm_mean=0.0, norm_var=1.0):
    st = net.struct()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_gmlp_value(C.byref(st), x.data_ptr(), x.shape[1], x.shape[0], _ptr(group_off), _ptr(perm),
                                            int(split), float(norm_mean), float(norm_var), out.data_ptr(),
                                            _stream(x.device)), "em_gmlp_value")
    return out


def actor_grad(net: GroupedMLP, x, action, old_logp, adv, moments, grad_out, loss3_out, slab, group_off=None, perm=None,
               split=1, clip=0.2, entropy_coeff=0.01):
    st = net.struct()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_gmlp_actor_grad(C.byref(st), net.log_std_ptr(), x.data_ptr(), x.shape[1], action.data_ptr(),
                                                 old_logp.data_ptr(), adv.data_ptr(), moments.data_ptr(), x.shape[0],
                                                 _ptr(group_off), _ptr(perm), int(split), float(clip), float(entropy_coeff),
                                                 slab.data_ptr(), grad_out.data_ptr(), loss3_out.data_ptr(),
                                                 _stream(x.device)), "em_gmlp_actor_grad")


def critic_grad(net: GroupedMLP, x, ret, moments, grad_out, loss3_out, slab, group_off=None, perm=None, split=1,
                value_loss_coeff=0.5, norm_mean=0.0, norm_var=1.0):
    st = net.struct()
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_gmlp_critic_grad(C.byref(st), x.data_ptr(), x.shape[1], ret.data_ptr(), moments.data_ptr(),
                                                  x.shape[0], _ptr(group_off), _ptr(perm), int(split), float(value_loss_coeff),
                                                  float(norm_mean), float(norm_var), slab.data_ptr(), grad_out.data_ptr(),
                                                  loss3_out.data_ptr(), _stream(x.device)), "em_gmlp_critic_grad")


def group_adam(param, grad, exp_avg, exp_avg_sq, step, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, max_grad_norm=0.5,
               norms_out=None):
    G, n_per = param.shape
    with torch.cuda.device(param.device):
        _lib.check(_lib.lib().em_group_adam(param.data_ptr(), grad.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(),
                                            n_per, G, float(lr), int(step), float(betas[0]), float(betas[1]), float(eps),
                                            float(max_grad_norm), _ptr(norms_out), _stream(param.device)), "em_group_adam")


def group_moments(adv, group_off, perm, n_groups, out):
    with torch.cuda.device(adv.device):
        _lib.check(_lib.lib().em_group_moments(adv.data_ptr(), group_off.data_ptr(), _ptr(perm), int(n_groups),
                                               out.data_ptr(), _stream(adv.device)), "em_group_moments")
    return out
