"""Independent PPO ("IPPO") and shared-actor PPO on the scaled topology (BASELINE.json configs[4]: 64 edge servers x
256 decider agents), on the grouped MLP kernels of ``csrc/gmlp.hip``.

The reference trains IPPO through RLlib's PPO with one policy per agent (``MRRL.PY:611-638``; ``ray[rllib]`` is not
vendored): **parity with the reference is unpinned**.  The build defines the independent learner natively: every
decider owns an actor (S+4 -> 64 -> 64 -> S+1, state-independent log_std) and a critic on its local observation
(S+4 -> 64 -> 64 -> 1) and runs the reference's own MAPPO arithmetic on its own transitions: Gaussian policy and action
formatting (``mappo_algorithm.py:62-71``, ``mappo_trainer.py:125-158``), GAE (``:241-262``) over the agent's own chain per
environment, advantage normalisation over the learner's batch (``:293``), PPO-clip / value / entropy losses
(``:302-320``), separate gradient clipping and Adam per network (``:328-332``).  ``shared=True`` replaces the 256 networks
by one actor and one local critic shared by all deciders (parameter sharing); MAPPO proper (shared actor + central
17,408-wide critic) is ``marl_amd/mappo_scaled.py``, reached through ``IPPOTrainer(algorithm="mappo")``.

Data parallel: environments are sharded over ranks; per update one all-reduce of the per-learner advantage moments
``[G, 3]`` and per epoch one all-reduce of the flat gradient buffer ``[G, n_actor + n_critic (+ loss sums)]`` (RCCL)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, gmlp
from .config import SCALED_ENV_CONFIG, mean_arrivals_bound, summarise_episode_metrics
from .env import METRIC_KEYS
from .env_scaled import PENDING_CAP, ScaledEdgeSimGym
from .mappo import _require_finite
from .parallel import GradSync

_STAT_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain")


def _stream(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class ScaledRolloutStorage:
    """Device buffers of ``ems_rollout``."""

    def __init__(self, n_envs, n_servers, n_deciders, horizon, device, capacity_entries=None, central=False,
                 entries_per_step=4.0):
        """``entries_per_step``: bound on the mean number of acting deciders per environment step used to size large
        batches (``marl_amd.config.mean_arrivals_bound``); small batches get the exact worst case."""
        B, S, D, T = int(n_envs), int(n_servers), int(n_deciders), int(horizon)
        if capacity_entries is None:
            worst = B * T * min(D, PENDING_CAP)
            capacity_entries = worst if worst <= (1 << 22) else int(B * T * min(D, PENDING_CAP, entries_per_step)) + B * PENDING_CAP
        self.n_envs, self.n_servers, self.n_deciders, self.horizon, self.device = B, S, D, T, device
        self.capacity_entries = E = int(capacity_entries)
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        self.t = dict(
            ent_obs=z((E, S + 4), f32), ent_action=z((E, S + 1), f32), ent_logp=z(E, f32), ent_value=z(E, f32),
            ent_env=z(E, i32), ent_agent=z(E, i32), ent_dst=z(E, i32), ent_adv=z(E, f32), ent_ret=z(E, f32),
            reward_tb=z((T, B), torch.float64), done_tb=z((T, B), u8), k_tb=z((T, B), u8), first_tb=z((T, B), i32),
            counters=z(T + 1, torch.int64), group_off=z(D + 3, torch.int64), perm=z(B * PENDING_CAP, i32), err_flag=z(1, i32))
        if central:        # MAPPO's central critic: state row of every entry and dense per-(step, env) values
            self.t.update(ent_state=z(E, i32), value_tb=z((T, B), f32))
        ptrs = [v.data_ptr() for v in self.t.values()] + ([] if central else [None, None])
        self.index_scratch = z((((B + 255) // 256 + 2 * D + 1 + 31) // 32) * 32 + 64 * D, i32)   # EMS_INDEX_SCRATCH_INTS
        self.struct = _lib.EmsRollout(B, E, S, D, T, 0, *ptrs, self.index_scratch.data_ptr())
        self.steps_recorded = 0

    def clear(self):
        self.t["counters"].zero_()
        self.t["k_tb"].zero_()
        self.t["err_flag"].zero_()        # sticky per batch, not per storage
        self.steps_recorded = 0
        self.values_through = 0           # recorded steps whose entries have a value (deferred critic, IPPO.compute_values)
        self.group_cache = None           # ((entries, steps), offsets, permutation) of IPPO._grouping

    def total_raw(self):
        """(entries recorded so far, error flag), without raising; synchronises."""
        return int(self.t["counters"][self.steps_recorded].item()), int(self.t["err_flag"].item())

    def overflow_message(self, entries):
        return (f"rollout storage capacity exceeded after {entries} entries were recorded (capacity {self.capacity_entries} "
                f"for {self.n_envs} envs x {self.horizon} steps): construct the trainer with a larger capacity_entries")

    def total(self) -> int:
        """Entries recorded so far; synchronises."""
        e, err = self.total_raw()
        if err != 0:
            raise RuntimeError(self.overflow_message(e))
        return e


class IPPO:
    def __init__(self, n_servers, n_deciders, lr=3e-4, gamma=0.99, gae_lambda=0.95, clip_param=0.2, value_loss_coeff=0.5,
                 entropy_coeff=0.01, max_grad_norm=0.5, num_sgd_iter=4, device="cuda:0", n_envs=1, horizon=1000,
                 capacity_entries=None, seed=0, shared=False, process_group=None, entries_per_step=4.0):
        self.S, self.D = int(n_servers), int(n_deciders)
        self.G = 1 if shared else self.D
        self.gamma, self.gae_lambda, self.clip_param = gamma, gae_lambda, clip_param
        self.value_loss_coeff, self.entropy_coeff = value_loss_coeff, entropy_coeff
        self.max_grad_norm, self.num_sgd_iter, self.lr = max_grad_norm, num_sgd_iter, lr
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("marl_amd.IPPO runs on an MI355X (device='cuda:N'); there is no CPU path")
        self._L = _lib.lib()
        self.sync = GradSync(process_group)
        self.actor = gmlp.GroupedMLP(self.G, self.S + 4, self.S + 1, True, self.device, seed=2 * seed)
        self.critic = gmlp.GroupedMLP(self.G, self.S + 4, 1, False, self.device, seed=2 * seed + 1)
        na, nc = self.actor.n_per, self.critic.n_per
        # one gradient buffer for the data-parallel all-reduce: [G, actor | actor loss3 | critic | critic loss3]
        self.grads = torch.zeros((self.G, na + 3 + nc + 3), dtype=torch.float32, device=self.device)
        self.g_actor = torch.zeros((self.G, na), dtype=torch.float32, device=self.device)
        self.g_critic = torch.zeros((self.G, nc), dtype=torch.float32, device=self.device)
        self.l_actor = torch.zeros((self.G, 3), dtype=torch.float32, device=self.device)
        self.l_critic = torch.zeros((self.G, 3), dtype=torch.float32, device=self.device)
        self.m_a, self.v_a = torch.zeros_like(self.actor.flat), torch.zeros_like(self.actor.flat)
        self.m_c, self.v_c = torch.zeros_like(self.critic.flat), torch.zeros_like(self.critic.flat)
        self.step_count = 0
        self.entries_per_step = float(entries_per_step)
        self.buffer = ScaledRolloutStorage(n_envs, self.S, self.D, horizon, self.device, capacity_entries,
                                           entries_per_step=entries_per_step)
        self.moments = torch.zeros((self.G, 3), dtype=torch.float64, device=self.device)
        self._chain = torch.zeros((int(n_envs), self.D, 2), dtype=torch.float64, device=self.device)
        # workgroups per network: independent learners map one network per CU; a shared network uses the whole chip
        self.split = 1 if self.G >= 64 else max(1, 256 // self.G)
        self.fwd_split = self.split          # workgroups per network in the rollout forwards (rows are independent)
        self._slab_a = self._slab_c = None
        self.last_update = {}
        # the value forward of a step feeds nothing before GAE: during a recorded rollout the trainer lets it run on a
        # side stream beside the policy forward and the environment step (overlap_value); join_value() orders the main
        # stream behind it again (called before the next step's index overwrites group_off/perm, and by update())
        self.overlap_value = False
        self.defer_values = True          # values of a recorded rollout: one launch over all entries (compute_values)
        self._side = self._ev_idx = self._ev_val = None
        self._value_pending = False

    def _value_async(self, fn):
        main = torch.cuda.current_stream(self.device)
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
            self._ev_idx, self._ev_val = torch.cuda.Event(), torch.cuda.Event()
        self._ev_idx.record(main)                       # everything queued so far: this step's index kernels
        self._side.wait_event(self._ev_idx)
        with torch.cuda.stream(self._side):
            fn()
            self._ev_val.record(self._side)
        self._value_pending = True

    def join_value(self):
        if self._value_pending:
            torch.cuda.current_stream(self.device).wait_event(self._ev_val)
            self._value_pending = False

    # -- rollout: MAPPOTrainer.train_episode steps 199-219 for all environments ----------------------------------------
    def record_and_act(self, t, env: ScaledEdgeSimGym, placement, resource, explore=True, noise=None, seed=0, episode=0,
                       env_offset=0, rng_step=None):
        ro = self.buffer
        tt = ro.t
        self.join_value()
        defer = self.defer_values and ro.horizon > 1
        # deferred critic: nothing reads the step's observation copy before the update, so the policy launch makes it while it
        # reads the rows from the environment's lists (no copy in the index launch: 10 -> 7 us per step)
        gather = defer and getattr(self, "gather_obs", True)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ems_rollout_index(C.byref(ro.struct), t, self.G, env.n_pending.data_ptr(),
                                                 env.pend_decider.data_ptr(), None if gather else env.pend_obs.data_ptr(),
                                                 _stream(self.device)), "ems_rollout_index")
        value = lambda: gmlp.value(self.critic, tt["ent_obs"], tt["ent_value"], tt["group_off"], tt["perm"], self.fwd_split)  # noqa: E731
        if defer:
            value = lambda: None  # noqa: E731  (update() evaluates every recorded entry in one launch before GAE)
        elif self.overlap_value:
            self._value_async(value)
        gmlp.policy(self.actor, tt["ent_obs"], tt["ent_env"], tt["ent_agent"], tt["ent_dst"], placement, resource,
                    tt["ent_action"], tt["ent_logp"], tt["group_off"], tt["perm"], self.fwd_split, noise=noise, explore=explore,
                    seed=seed, episode=episode, rng_step=t if rng_step is None else rng_step, env_offset=env_offset,
                    x_src=env.pend_obs if gather else None)
        if defer or not self.overlap_value:
            value()
        ro.steps_recorded = t + 1

    def _grouping(self, E):
        """Entries [0, E) grouped by learner: (offsets, permutation), stable (fixed summation order per learner); cached
        for the (E, steps recorded) it was built for."""
        ro = self.buffer
        key = (E, ro.steps_recorded)
        if getattr(ro, "group_cache", None) is None or ro.group_cache[0] != key:
            dev = self.device
            if self.G > 1:
                off, perm = gmlp.group_index(self.buffer.t["ent_agent"][:E], self.G)
            else:
                off, perm = torch.tensor([0, E], dtype=torch.int64, device=dev), None
            ro.group_cache = (key, off, perm)
        return ro.group_cache[1], ro.group_cache[2]

    def compute_values(self):
        """Deferred mode: the critics' forward over every recorded entry that has no value yet, in one launch (the rollout
        needs no value before GAE, and the per-step value launch could never share a CU with the policy launch: their
        registers do not fit together).  A no-op otherwise."""
        ro = self.buffer
        if not (self.defer_values and ro.horizon > 1) or getattr(ro, "values_through", 0) == ro.steps_recorded:
            return
        E = ro.total()
        if E > 0:
            off, perm = self._grouping(E)
            gmlp.value(self.critic, ro.t["ent_obs"][:E], ro.t["ent_value"], off, perm, self.split)
        ro.values_through = ro.steps_recorded

    # -- update ----------------------------------------------------------------------------------------------------------
    def update(self):
        ro = self.buffer
        self.join_value()
        E = self._global_gate()
        if E is None:
            return {}
        tt, L, dev = ro.t, self._L, self.device
        x = tt["ent_obs"][:E]
        off = perm = None
        if E > 0:
            off, perm = self._grouping(E)
            self.compute_values()
            with torch.cuda.device(dev):
                _lib.check(L.em_gae_agents(ro.n_envs, ro.steps_recorded, self.D, tt["reward_tb"].data_ptr(), tt["done_tb"].data_ptr(),
                                           tt["k_tb"].data_ptr(), tt["first_tb"].data_ptr(), tt["ent_agent"].data_ptr(),
                                           tt["ent_value"].data_ptr(), float(self.gamma), float(self.gae_lambda),
                                           self._chain.data_ptr(), tt["ent_adv"].data_ptr(), tt["ent_ret"].data_ptr(), _stream(dev)),
                           "em_gae_agents")
            gmlp.group_moments(tt["ent_adv"], off, perm, self.G, self.moments)
        else:                                # empty shard: zero moments, zero gradients, every collective still joined
            self.moments.zero_()
        n_global = self._sum_moments()       # per-learner moments, summed over ranks
        na, nc = self.actor.n_per, self.critic.n_per
        sa, sc = self.actor.slab_floats(), self.critic.slab_floats()
        rows = self.G * 4 * self.split
        if self._slab_a is None:
            self._slab_a = torch.empty(rows * sa, dtype=torch.float32, device=dev)
            self._slab_c = torch.empty(rows * sc, dtype=torch.float32, device=dev)
        acc = torch.zeros(6, dtype=torch.float64, device=dev)
        for _ in range(int(self.num_sgd_iter)):
            if E > 0:
                gmlp.actor_grad(self.actor, x, tt["ent_action"], tt["ent_logp"], tt["ent_adv"], self.moments, self.g_actor,
                                self.l_actor, self._slab_a, off, perm, self.split, self.clip_param, self.entropy_coeff)
                gmlp.critic_grad(self.critic, x, tt["ent_ret"], self.moments, self.g_critic, self.l_critic, self._slab_c, off,
                                 perm, self.split, self.value_loss_coeff)
            else:
                for buf in (self.g_actor, self.l_actor, self.g_critic, self.l_critic):
                    buf.zero_()
            if self.sync.active:   # the one data-path collective per epoch
                g = self.grads
                g[:, :na] = self.g_actor; g[:, na:na + 3] = self.l_actor
                g[:, na + 3:na + 3 + nc] = self.g_critic; g[:, na + 3 + nc:] = self.l_critic
                self.sync.sum_grads(g)
                self.g_actor.copy_(g[:, :na]); self.l_actor.copy_(g[:, na:na + 3])
                self.g_critic.copy_(g[:, na + 3:na + 3 + nc]); self.l_critic.copy_(g[:, na + 3 + nc:])
            self.step_count += 1
            gmlp.group_adam(self.actor.flat, self.g_actor, self.m_a, self.v_a, self.step_count, lr=self.lr,
                            max_grad_norm=self.max_grad_norm)
            gmlp.group_adam(self.critic.flat, self.g_critic, self.m_c, self.v_c, self.step_count, lr=self.lr,
                            max_grad_norm=self.max_grad_norm)
            acc[:3] += self.l_actor.sum(0).double()
            acc[3:] += self.l_critic.sum(0).double()
        ro.clear()
        s = acc.cpu()
        _require_finite(s)
        k = float(self.num_sgd_iter) * n_global
        self.last_update = {"actor_loss": float(-s[0] / k), "critic_loss": float(0.5 * s[3] / k), "entropy": float(s[1] / k),
                            "transitions": n_global}
        return dict(self.last_update)

    def _global_gate(self):
        """Entries of this rank, or None when NO rank recorded anything.  The decisions that would leave update() before a
        collective are taken on global totals: an overflow on one rank raises on all, a rank with an empty shard goes on
        (its kernels see zero rows, its gradient slabs reduce to zeros) and joins every all-reduce."""
        ro = self.buffer
        E, err = ro.total_raw() if ro.steps_recorded else (0, 0)
        E_g, err_g = self.sync.sum_counts([E, err], self.device)
        if err_g:
            msg = ro.overflow_message(E) if err else "rollout storage capacity exceeded on another rank"
            ro.clear()
            raise RuntimeError(msg)
        if E_g == 0:
            ro.clear()
            return None
        return E

    def _sum_moments(self) -> int:
        self.sync.sum_grads(self.moments)          # plain SUM over ranks of the [G, 3] float64 moments
        return int(round(float(self.moments[:, 0].sum().item())))

    # -- checkpoint: flat parameter and Adam buffers --------------------------------------------------------------------
    def save(self, path):
        torch.save({"actor": self.actor.flat.cpu(), "critic": self.critic.flat.cpu(), "m_a": self.m_a.cpu(), "v_a": self.v_a.cpu(),
                    "m_c": self.m_c.cpu(), "v_c": self.v_c.cpu(), "step": self.step_count,
                    "shape": (self.S, self.D, self.G)}, path)

    def load(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        if tuple(ck["shape"]) != (self.S, self.D, self.G):
            raise ValueError("checkpoint was written for another topology")
        for dst, key in ((self.actor.flat, "actor"), (self.critic.flat, "critic"), (self.m_a, "m_a"), (self.v_a, "v_a"),
                         (self.m_c, "m_c"), (self.v_c, "v_c")):
            dst.copy_(ck[key].to(self.device))
        self.step_count = int(ck["step"])


class IPPOTrainer:
    """The MAPPOTrainer interface (mappo_trainer.py:23-394: ``train_episode``, ``evaluate``, ``train``) for independent
    learners on the scaled topology; one call of ``train_episode`` plays one episode in every environment."""

    def __init__(self, env_config=None, seed=0, device="cuda:0", n_envs=1, rank=0, world_size=1, process_group=None,
                 shared=False, capacity_entries=None, verbose=True, algorithm="ippo"):
        """``algorithm``: "ippo" (one actor and one local critic per decider), "shared" (= ``shared=True``: one actor and one
        local critic for all deciders) or "mappo" (shared actor + central critic, marl_amd/mappo_scaled.py)."""
        if shared:
            algorithm = "shared"
        self.device = torch.device("cuda:0" if str(device) == "cuda" else device)
        self.seed, self.n_envs, self.rank, self.world_size = int(seed), int(n_envs), int(rank), int(world_size)
        cfg = dict(SCALED_ENV_CONFIG if env_config is None else env_config)
        self.env_offset = self.rank * self.n_envs
        self.env = ScaledEdgeSimGym(cfg, n_envs=self.n_envs, device=self.device, seed=self.seed, env_offset=self.env_offset)
        self.S, self.D = self.env.n_nodes, self.env.n_deciders
        self.algorithm = algorithm
        eps_bound = mean_arrivals_bound(cfg)
        if algorithm == "mappo":
            from .mappo_scaled import ScaledMAPPO
            self.algo = ScaledMAPPO(self.S, self.D, device=self.device, n_envs=self.n_envs, horizon=self.env.episode_len,
                                    capacity_entries=capacity_entries, seed=self.seed, process_group=process_group,
                                    entries_per_step=eps_bound)
        elif algorithm in ("ippo", "shared"):
            self.algo = IPPO(self.S, self.D, device=self.device, n_envs=self.n_envs, horizon=self.env.episode_len,
                             capacity_entries=capacity_entries, seed=self.seed, shared=algorithm == "shared",
                             process_group=process_group, entries_per_step=eps_bound)
        else:
            raise ValueError(f"unknown algorithm {algorithm!r}")
        self.placement = torch.zeros((self.n_envs, PENDING_CAP), dtype=torch.int32, device=self.device)
        self.resource = torch.full((self.n_envs, PENDING_CAP), 0.5, dtype=torch.float32, device=self.device)
        self._train_storage = self.algo.buffer
        self._scratch = None
        self.policy_episode = 0
        # value forward on a side stream during recorded rollouts (identical results): pays once the step is GPU bound
        # (130 -> 123 ms per 1000 steps at 16,384 envs); below that the two extra stream calls cost more than they hide
        self.overlap_value = self.n_envs >= 8192
        if verbose and self.rank == 0:
            kind = {"shared": "shared-parameter PPO", "mappo": "MAPPO (shared actor, central critic)"}.get(
                algorithm, f"IPPO ({self.D} independent learners)")
            print(f"[OK] {kind} (MI355X) initialised: {self.S} servers, {self.D} deciders, envs/GPU {self.n_envs}")

    def train_episode(self, explore=True, record_buffer=True, noise=None):
        env, a = self.env, self.algo
        T = env.episode_len
        if record_buffer:
            a.buffer = self._train_storage
            a.buffer.clear()
        else:
            if self._scratch is None:
                self._scratch = ScaledRolloutStorage(self.n_envs, self.S, self.D, 1, self.device,
                                                     capacity_entries=self.n_envs * PENDING_CAP,
                                                     central=self.algorithm == "mappo")
            a.buffer = self._scratch
        ro = a.buffer
        env.reset()
        self.policy_episode += 1
        total_reward = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)
        a.overlap_value = bool(record_buffer) and self.overlap_value
        try:
            for t in range(T):
                slot = t if record_buffer else 0
                if not record_buffer:
                    ro.t["counters"].zero_()
                a.record_and_act(slot, env, self.placement, self.resource, explore=explore, noise=noise, seed=self.seed,
                                 episode=self.policy_episode, env_offset=self.env_offset, rng_step=t)
                if record_buffer:
                    env.step(self.placement, self.resource, by_slot=True, reward_seen_out=ro.t["reward_tb"][slot],
                             done_out=ro.t["done_tb"][slot])
                else:
                    env.step(self.placement, self.resource, by_slot=True)
                    total_reward += env.reward
        finally:
            a.overlap_value = False
            a.join_value()
            a.buffer = self._train_storage
        if record_buffer:
            total_reward = ro.t["reward_tb"][:T].sum(dim=0)
            if hasattr(a, "compute_values"):
                a.compute_values()
        metrics = env.episode_metrics()
        mean, p99_missing = summarise_episode_metrics(metrics, env.n_pending > 0)      # mappo_trainer.py:222-231
        idx = {k: i for i, k in enumerate(METRIC_KEYS)}
        stats = {"episode_reward": float(total_reward.mean().item()), "episode_steps": T}
        for k in _STAT_KEYS:
            stats[k] = float(mean[idx[k]])
        stats["p99_envs_missing"] = p99_missing
        stats["per_env"] = {"episode_reward": total_reward, "metrics": metrics}
        return stats

    def evaluate(self, num_episodes=100):
        prev = self.env.disable_env_exploration
        self.env.disable_env_exploration = True
        acc = {k: [] for k in _STAT_KEYS}
        try:
            for _ in range(max(1, -(-int(num_episodes) // self.n_envs))):
                stats = self.train_episode(explore=False, record_buffer=False)
                for k in _STAT_KEYS:
                    acc[k].append(stats[k])
        finally:
            self.env.disable_env_exploration = prev
        return {k: float(np.mean(v)) for k, v in acc.items()}

    def train(self, num_iterations=10, eval_interval=10, eval_episodes=10):
        log = []
        for it in range(1, num_iterations + 1):
            st = self.train_episode(explore=True, record_buffer=True)
            up = self.algo.update()
            entry = {"algorithm": self.algorithm.upper(), "seed": self.seed, "iteration": it, "episode_return_mean": st["episode_reward"], **up}
            if it % eval_interval == 0:
                entry.update(self.evaluate(eval_episodes))
            log.append(entry)
        return log
