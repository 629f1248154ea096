"""MAPPO on the scaled topology (BASELINE.json configs[4]: 64 servers x 256 deciders): the reference's shared actor
(``MAPPOActor``, mappo_algorithm.py:16-71) and its CENTRAL critic (``MAPPOCritic``, :74-90) on the concatenated state of
``MAPPOTrainer.obs_dict_to_state`` (mappo_trainer.py:101-123), which is 256 x 68 = 17,408 wide here and >99 % zeros.  The
critic's first layer is therefore evaluated sparsely over the deciders that hold a task (``csrc/central_critic.hip``);
everything else is the reference's MAPPO: one value per (environment, step) shared by that step's transitions, GAE over
each environment's flattened transition list (mappo_algorithm.py:241-262, ``em_gae``), advantages normalised over the
whole buffer (:293), full-batch PPO epochs (:300-332).  Parity: pinned to the torch restatement of the dense network
(tests/test_mappo_scaled_gpu.py); the topology itself is the build's generalisation."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib, gmlp
from .ippo import IPPO, ScaledRolloutStorage, _stream
from .mappo import _require_finite

HID = 64


class CentralCritic:
    """Flat parameters ``[w1 (64 x D*W) | b1 | w2 | b2 | w3 | b3]`` in torch.nn.Linear layout and initialisation."""

    def __init__(self, n_deciders, width, device, seed=0):
        self.D, self.W = int(n_deciders), int(width)
        self.state_dim = self.D * self.W
        sizes = [("w1", HID * self.state_dim, self.state_dim), ("b1", HID, self.state_dim), ("w2", HID * HID, HID), ("b2", HID, HID),
                 ("w3", HID, HID), ("b3", 1, HID)]
        gen = torch.Generator().manual_seed(int(seed))
        self.offsets, off, parts = {}, 0, []
        for name, n, fan_in in sizes:
            self.offsets[name] = (off, n)
            off += n
            parts.append((torch.rand(n, generator=gen) * 2 - 1) / math.sqrt(fan_in))
        self.n_per = off
        self.flat = torch.cat(parts).to(device).contiguous().view(1, -1)       # [1, n]: the layout group_adam expects

    def struct(self) -> _lib.EmCentralCritic:
        base = self.flat.data_ptr()
        p = {k: base + 4 * o for k, (o, _) in self.offsets.items()}
        return _lib.EmCentralCritic(p["w1"], p["b1"], p["w2"], p["b2"], p["w3"], p["b3"], self.D, self.W, self.state_dim, 0)


class ScaledMAPPO(IPPO):
    """Shared actor + central critic.  Same surface as :class:`marl_amd.ippo.IPPO`."""

    def __init__(self, n_servers, n_deciders, device="cuda:0", n_envs=1, horizon=1000, capacity_entries=None, seed=0,
                 process_group=None, **hyper):
        super().__init__(n_servers, n_deciders, device=device, n_envs=n_envs, horizon=horizon, capacity_entries=1, seed=seed,
                         shared=True, process_group=process_group, **hyper)
        dev = self.device
        self.buffer = ScaledRolloutStorage(n_envs, self.S, self.D, horizon, dev, capacity_entries, central=True,
                                           entries_per_step=self.entries_per_step)
        self.critic = CentralCritic(self.D, self.S + 4, dev, seed=2 * seed + 1)
        nc = self.critic.n_per
        self.g_critic = torch.zeros((1, nc), dtype=torch.float32, device=dev)
        self.m_c, self.v_c = torch.zeros_like(self.critic.flat), torch.zeros_like(self.critic.flat)
        self.grads = torch.zeros((1, self.actor.n_per + 3 + nc + 3), dtype=torch.float32, device=dev)
        # workgroups per decider block of the sparse first layer: two per CU (their waits for the scattered rows overlap;
        # one took 257 instead of 227 ms per update at 256 deciders)
        self.cc_split = max(1, 512 // self.D)
        self.moments = torch.zeros(3, dtype=torch.float64, device=dev)
        self._gae_partials = torch.zeros(3 * ((int(n_envs) + 255) // 256), dtype=torch.float64, device=dev)
        self._partial = self._dz = self._cc_slab = None

    def _partials(self, n_entries):
        if self._partial is None or self._partial.shape[0] < n_entries:
            self._partial = torch.empty((n_entries, HID), dtype=torch.float32, device=self.device)
        return self._partial

    def record_and_act(self, t, env, placement, resource, explore=True, noise=None, seed=0, episode=0, env_offset=0,
                       rng_step=None):
        ro = self.buffer
        tt, dev, D = ro.t, self.device, self.D
        self.join_value()
        defer = self.defer_values and ro.horizon > 1    # one launch over all recorded states at the end (compute_values)
        # the per-decider grouping of a step serves the central critic's sparse first layer only: deferred, the step is
        # indexed as one group (no histogram, no cursor atomics, identity permutation)
        ng = 1 if defer else D
        gather = defer and getattr(self, "gather_obs", True)     # as IPPO.record_and_act: the policy launch copies the rows
        with torch.cuda.device(dev):
            _lib.check(self._L.ems_rollout_index(C.byref(ro.struct), t, ng, env.n_pending.data_ptr(), env.pend_decider.data_ptr(),
                                                 None if gather else env.pend_obs.data_ptr(), _stream(dev)), "ems_rollout_index")
        cc = self.critic.struct()
        B = ro.n_envs
        part = self._partials(ro.capacity_entries)

        def value():
            with torch.cuda.device(dev):
                _lib.check(self._L.em_cc_forward(C.byref(cc), tt["ent_obs"].data_ptr(), tt["group_off"].data_ptr(), tt["perm"].data_ptr(),
                                                 self.cc_split, part.data_ptr(), B, tt["k_tb"][t].data_ptr(),
                                                 tt["first_tb"][t].data_ptr(), 0.0, 1.0, tt["value_tb"][t].data_ptr(), _stream(dev)),
                           "em_cc_forward")

        if defer:
            value = lambda: None  # noqa: E731
        elif self.overlap_value:
            self._value_async(value)
        # shared actor: the step as one group ({0, entries} sits behind the per-decider offsets), same permutation
        gmlp.policy(self.actor, tt["ent_obs"], tt["ent_env"], tt["ent_agent"], tt["ent_dst"], placement, resource,
                    tt["ent_action"], tt["ent_logp"], tt["group_off"][(D + 1 if ng > 1 else 0):], tt["perm"], self.split, noise=noise, explore=explore,
                    seed=seed, episode=episode, rng_step=t if rng_step is None else rng_step, env_offset=env_offset,
                    x_src=env.pend_obs if gather else None)
        if defer or not self.overlap_value:
            value()
        ro.steps_recorded = t + 1

    def _decider_grouping(self, E):
        """Entries [0, E) grouped by decider (stable: fixed summation order per decider block of the first layer); cached for
        the (E, steps recorded) it was built for and shared by compute_values() and update()."""
        ro = self.buffer
        key = ("decider", E, ro.steps_recorded)
        if getattr(ro, "group_cache", None) is None or ro.group_cache[0] != key:
            off, perm = gmlp.group_index(ro.t["ent_agent"][:E], self.D)
            ro.group_cache = (key, off, perm)
        return ro.group_cache[1], ro.group_cache[2]

    def compute_values(self):
        """Deferred mode: the central critic over every recorded state that has no value yet, in one em_cc_forward (nothing
        in the rollout reads a value before GAE; per step the sparse first layer ran on a side stream beside the policy
        launch, 61 us a step).  A no-op otherwise."""
        ro = self.buffer
        if not (self.defer_values and ro.horizon > 1) or getattr(ro, "values_through", 0) == ro.steps_recorded:
            return
        E = ro.total()
        tt, dev, T, B = ro.t, self.device, ro.steps_recorded, ro.n_envs
        if E > 0:
            off, perm = self._decider_grouping(E)
            cc = self.critic.struct()
            part = self._partials(E)
            with torch.cuda.device(dev):
                _lib.check(self._L.em_cc_forward(C.byref(cc), tt["ent_obs"].data_ptr(), off.data_ptr(), perm.data_ptr(),
                                                 self.cc_split, part.data_ptr(), T * B, tt["k_tb"].data_ptr(),
                                                 tt["first_tb"].data_ptr(), 0.0, 1.0, tt["value_tb"].data_ptr(), _stream(dev)),
                           "em_cc_forward")
        ro.values_through = ro.steps_recorded

    def update(self):
        ro = self.buffer
        self.join_value()
        E = self._global_gate()
        if E is None:
            return {}
        tt, L, dev, T, B = ro.t, self._L, self.device, ro.steps_recorded, ro.n_envs
        self.compute_values()
        # MAPPO.compute_gae over each environment's flattened list (em_gae reads the dense [T][B] arrays)
        g = _lib.EmRollout()
        g.n_envs, g.horizon, g.n_nodes = B, T, self.S
        g.reward_tb, g.value_tb, g.done_tb = tt["reward_tb"].data_ptr(), tt["value_tb"].data_ptr(), tt["done_tb"].data_ptr()
        g.k_tb, g.first_tb, g.ent_adv, g.ent_ret = tt["k_tb"].data_ptr(), tt["first_tb"].data_ptr(), tt["ent_adv"].data_ptr(), tt["ent_ret"].data_ptr()
        with torch.cuda.device(dev):
            _lib.check(L.em_gae(C.byref(g), float(self.gamma), float(self.gae_lambda), self._gae_partials.data_ptr(),
                                self._gae_partials.numel(), self.moments.data_ptr(), _stream(dev)), "em_gae")
        n_global = self.sync.sum_moments(self.moments)
        x = tt["ent_obs"][:E]
        off_a = torch.tensor([0, E], dtype=torch.int64, device=dev)
        if E > 0:
            off_c, perm_c = self._decider_grouping(E)
        cc = self.critic.struct()
        n_states = T * B
        if self._dz is None or self._dz.shape[0] < n_states:
            self._dz = torch.empty((n_states, HID), dtype=torch.float32, device=dev)
        need = int(L.em_cc_scratch_floats(C.byref(cc), self.cc_split, n_states))
        if self._cc_slab is None or self._cc_slab.numel() < need:
            self._cc_slab = torch.empty(need, dtype=torch.float32, device=dev)
        if self._slab_a is None:
            self._slab_a = torch.empty(4 * self.split * self.actor.slab_floats(), dtype=torch.float32, device=dev)
        na, nc = self.actor.n_per, self.critic.n_per
        mom = self.moments.view(1, 3)
        acc = torch.zeros(6, dtype=torch.float64, device=dev)
        part = self._partials(max(E, 1))
        for _ in range(int(self.num_sgd_iter)):
            if E == 0:                       # empty shard: zero contributions, every collective still joined
                for buf in (self.g_actor, self.l_actor, self.g_critic, self.l_critic):
                    buf.zero_()
            else:
                gmlp.actor_grad(self.actor, x, tt["ent_action"], tt["ent_logp"], tt["ent_adv"], mom, self.g_actor, self.l_actor,
                                self._slab_a, off_a, None, self.split, self.clip_param, self.entropy_coeff)
                cc = self.critic.struct()
                with torch.cuda.device(dev):
                    _lib.check(L.em_cc_grad(C.byref(cc), x.data_ptr(), tt["ent_state"].data_ptr(), tt["ent_ret"].data_ptr(),
                                            off_c.data_ptr(), perm_c.data_ptr(), self.cc_split, part.data_ptr(), n_states,
                                            tt["k_tb"].data_ptr(), tt["first_tb"].data_ptr(), n_global,
                                            float(self.value_loss_coeff), 0.0, 1.0, self._dz.data_ptr(), self._cc_slab.data_ptr(),
                                            self.g_critic.data_ptr(), self.l_critic.data_ptr(), _stream(dev)), "em_cc_grad")
            if self.sync.active:
                gg = self.grads
                gg[:, :na] = self.g_actor; gg[:, na:na + 3] = self.l_actor
                gg[:, na + 3:na + 3 + nc] = self.g_critic; gg[:, na + 3 + nc:] = self.l_critic[:1]
                self.sync.sum_grads(gg)
                self.g_actor.copy_(gg[:, :na]); self.l_actor.copy_(gg[:, na:na + 3])
                self.g_critic.copy_(gg[:, na + 3:na + 3 + nc]); self.l_critic[:1].copy_(gg[:, na + 3 + nc:])
            self.step_count += 1
            gmlp.group_adam(self.actor.flat, self.g_actor, self.m_a, self.v_a, self.step_count, lr=self.lr,
                            max_grad_norm=self.max_grad_norm)
            gmlp.group_adam(self.critic.flat, self.g_critic, self.m_c, self.v_c, self.step_count, lr=self.lr,
                            max_grad_norm=self.max_grad_norm)
            acc[:3] += self.l_actor[0].double()
            acc[3:] += self.l_critic[0].double()
        ro.clear()
        s = acc.cpu()
        _require_finite(s)
        k = float(self.num_sgd_iter) * n_global
        self.last_update = {"actor_loss": float(-s[0] / k), "critic_loss": float(0.5 * s[3] / k), "entropy": float(s[1] / k),
                            "transitions": n_global}
        return dict(self.last_update)
