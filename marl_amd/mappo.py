"""MAPPO on MI355X: the reference's ``mappo_algorithm.MAPPO`` surface (constructor arguments, attributes poked
by ``run_mappo_batch.py:255-261``, ``update()``, ``save()/load()`` with the same checkpoint keys) over the HIP
kernels of ``csrc/mappo_fwd.hip`` / ``csrc/mappo_bwd.hip``.  torch holds the device buffers (parameters are
``nn.Parameter`` views of one flat float32 buffer so reference checkpoints load unchanged) and, for
multi-GPU, runs the one gradient all-reduce over RCCL; no torch autograd, optimizer or nn.functional op is on
the compute path."""
from __future__ import annotations

import ctypes as C
import math
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .parallel import GradSync

HID = 64


def _stream(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class _FlatNet(nn.Module):
    """Parameter container with the reference's module/parameter names; compute goes through the C ABI."""

    def _adopt(self, flat: torch.Tensor, names_shapes):
        off = 0
        self._slices = OrderedDict()
        for name, shape in names_shapes:
            n = int(torch.Size(shape).numel())
            view = flat[off:off + n].view(shape)
            mod_name, _, p_name = name.rpartition(".")
            holder = self.get_submodule(mod_name) if mod_name else self
            with torch.no_grad():
                view.copy_(getattr(holder, p_name).detach().to(flat.device))
            holder._parameters[p_name] = nn.Parameter(view, requires_grad=False)
            self._slices[name] = (off, n)
            off += n
        self.flat = flat[:off]
        return off

    def mlp3(self, w1, w2, w3) -> _lib.EmMlp3:
        return _lib.EmMlp3(w1.weight.data_ptr(), w1.bias.data_ptr(), w2.weight.data_ptr(), w2.bias.data_ptr(),
                           w3.weight.data_ptr(), w3.bias.data_ptr(), w1.weight.shape[1], HID, w3.weight.shape[0])


class MAPPOActor(_FlatNet):
    """Shared actor 14->64->64->11 + state-independent log_std (mappo_algorithm.py:16-71)."""

    def __init__(self, obs_dim, action_dim, hidden_dim=HID):
        super().__init__()
        if hidden_dim != HID:
            raise ValueError("the MI355X kernels are built for hidden_dim=64")
        self.fc1 = nn.Linear(obs_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.mu = nn.Linear(hidden_dim, action_dim)
        self.log_std = nn.Parameter(torch.zeros(action_dim))
        self.obs_dim, self.action_dim = obs_dim, action_dim

    def adopt(self, flat):
        return self._adopt(flat, [("fc1.weight", (HID, self.obs_dim)), ("fc1.bias", (HID,)), ("fc2.weight", (HID, HID)),
                                  ("fc2.bias", (HID,)), ("mu.weight", (self.action_dim, HID)), ("mu.bias", (self.action_dim,)),
                                  ("log_std", (self.action_dim,))])

    def net(self):
        return self.mlp3(self.fc1, self.fc2, self.mu)

    def forward(self, obs):
        return mlp3_forward(self.net(), obs)

    def get_action_log_prob(self, obs, action=None):
        """mappo_algorithm.py:50-71 on device tensors: ``(action, log_prob, entropy)`` with log-prob and entropy summed
        over the action components.  ``action=None`` samples a = mu + sigma * eps; eps is drawn exactly as
        ``Normal.rsample()`` draws it on a CPU module (``torch.normal(zeros, ones)`` on torch's global CPU generator), so
        ``torch.manual_seed`` reproduces the reference's samples."""
        obs = torch.as_tensor(obs, dtype=torch.float32, device=self.flat.device).reshape(-1, self.obs_dim).contiguous()
        mu = self.forward(obs)
        rows, A, dev = mu.shape[0], self.action_dim, mu.device
        L = _lib.lib()
        with torch.cuda.device(dev):
            s = _stream(dev)
            if action is None:
                eps = torch.normal(torch.zeros(rows, A), torch.ones(rows, A)).to(dev)
                action = torch.empty_like(mu)
                _lib.check(L.em_gauss_sample(mu.data_ptr(), self.log_std.data_ptr(), eps.data_ptr(), rows, A, action.data_ptr(), s),
                           "em_gauss_sample")
            else:
                action = torch.as_tensor(action, dtype=torch.float32, device=dev).reshape(rows, A).contiguous()
            logp = torch.empty(rows, dtype=torch.float32, device=dev)
            ent = torch.empty(rows, dtype=torch.float32, device=dev)
            _lib.check(L.em_gauss_logp(mu.data_ptr(), self.log_std.data_ptr(), action.data_ptr(), rows, A, logp.data_ptr(),
                                       ent.data_ptr(), s), "em_gauss_logp")
        return action, logp, ent


class MAPPOCritic(_FlatNet):
    """Central critic 140->64->64->1 (mappo_algorithm.py:74-90)."""

    def __init__(self, state_dim, hidden_dim=HID):
        super().__init__()
        if hidden_dim != HID:
            raise ValueError("the MI355X kernels are built for hidden_dim=64")
        self.fc1 = nn.Linear(state_dim, hidden_dim)
        self.fc2 = nn.Linear(hidden_dim, hidden_dim)
        self.value_out = nn.Linear(hidden_dim, 1)
        self.state_dim = state_dim

    def adopt(self, flat):
        return self._adopt(flat, [("fc1.weight", (HID, self.state_dim)), ("fc1.bias", (HID,)), ("fc2.weight", (HID, HID)),
                                  ("fc2.bias", (HID,)), ("value_out.weight", (1, HID)), ("value_out.bias", (1,))])

    def net(self):
        return self.mlp3(self.fc1, self.fc2, self.value_out)

    def forward(self, state):
        return mlp3_forward(self.net(), state)


class ValueNormalizer(nn.Module):
    """Buffers of the reference normaliser (mappo_algorithm.py:367-409).  Nothing in the reference ever calls
    update(), so it stays the map x -> x / sqrt(1 + 1e-8); the kernels take mean/var as scalars."""

    def __init__(self, input_shape, device="cpu"):
        super().__init__()
        self.register_buffer("running_mean", torch.zeros(input_shape, device=device))
        self.register_buffer("running_var", torch.ones(input_shape, device=device))
        self.register_buffer("count", torch.ones((), device=device))

    def forward(self, x):
        return (x - self.running_mean) / torch.sqrt(self.running_var + 1e-8)

    def denormalize(self, x):
        return x * torch.sqrt(self.running_var + 1e-8) + self.running_mean

    def update(self, x):
        self.update_from_moments(x.mean(dim=0), x.var(dim=0), x.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        """mappo_algorithm.py:396-409 (bookkeeping of three buffers; the kernels read them as scalars)."""
        delta = batch_mean - self.running_mean
        tot_count = self.count + batch_count
        m2 = self.running_var * self.count + batch_var * batch_count + delta ** 2 * self.count * batch_count / tot_count
        self.running_mean = self.running_mean + delta * batch_count / tot_count
        self.running_var = m2 / tot_count
        self.count = tot_count


def mlp3_forward(net: _lib.EmMlp3, x: torch.Tensor) -> torch.Tensor:
    x = x.contiguous().float()
    rows = x.shape[0]
    y = torch.empty((rows, net.out_dim), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().em_mlp3_forward(C.byref(net), x.data_ptr(), x.shape[1], rows, y.data_ptr(), _stream(x.device)),
                   "em_mlp3_forward")
    return y


class HipAdam:
    """State of torch.optim.Adam for one network on flat buffers; the update itself is em_adam_step.
    ``param_groups[0]["lr"]`` can be overridden as run_mappo_batch.py:255-256 does; state_dict()/load_state_dict()
    use torch.optim.Adam's layout (parameter order = module.parameters()) so checkpoints interchange."""

    def __init__(self, module: _FlatNet, lr):
        self.module = module
        self.param_groups = [{"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                              "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                              "fused": None, "decoupled_weight_decay": False}]
        self.exp_avg = torch.zeros_like(module.flat)
        self.exp_avg_sq = torch.zeros_like(module.flat)
        self.step_count = 0

    def _torch_order(self):
        return [name for name, _ in self.module.named_parameters()]

    def state_dict(self):
        names = self._torch_order()
        state = {}
        if self.step_count > 0:
            for i, name in enumerate(names):
                off, n = self.module._slices[name]
                shape = dict(self.module.named_parameters())[name].shape
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).clone()}
        group = dict(self.param_groups[0])
        group["params"] = list(range(len(names)))
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        names = self._torch_order()
        g = sd["param_groups"][0]
        for k in ("lr", "betas", "eps"):
            if k in g:
                self.param_groups[0][k] = g[k]
        self.step_count = 0
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for i, name in enumerate(names):
            st = sd["state"].get(i)
            if st is None:
                continue
            off, n = self.module._slices[name]
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(st["step"]))


class RolloutBuffer:
    """The reference's python-list buffer (mappo_algorithm.py:93-131), same members and methods: a host-side container for
    callers that feed MAPPO one dict transition at a time (``MAPPOTrainer._store_transition``).  ``MAPPO.update()`` uploads
    it into a one-environment ``RolloutStorage`` and runs the same device kernels as the batched path."""

    def __init__(self):
        self.clear()

    def add(self, obs, state, action, reward, done, log_prob, value):
        self.observations.append(obs)
        self.states.append(state)
        self.actions.append(action)
        self.rewards.append(reward)
        self.dones.append(done)
        self.log_probs.append(log_prob)
        self.values.append(value)

    def clear(self):
        self.observations, self.states, self.actions = [], [], []
        self.rewards, self.dones, self.log_probs, self.values = [], [], [], []

    def get(self):
        return {"observations": self.observations, "states": self.states, "actions": self.actions, "rewards": self.rewards,
                "dones": self.dones, "log_probs": self.log_probs, "values": self.values}

    def __len__(self):
        return len(self.rewards)


class RolloutStorage:
    """Device buffers of ``em_rollout`` for one batch of episodes (RolloutBuffer, mappo_algorithm.py:93-131).  The
    reference's ``add/get`` list interface is kept beside it (``staged``): transitions added one at a time are uploaded by
    ``MAPPO.update()``."""

    def __init__(self, n_envs, n_nodes, horizon, device, capacity_entries=None, capacity_steps=None, entries_per_step=2.2,
                 by_reference=False):
        """``entries_per_step``: bound on the MEAN number of acting agents per environment step used to size large
        batches (small ones get the exact worst case n_nodes); :func:`marl_amd.config.mean_arrivals_bound` derives it
        from the arrival-rate range and the time-step clip of an ENV_CONFIG.

        ``by_reference``: states are not copied.  The storage keeps the dense observation tensor of every step
        (``obs_tb`` [horizon + 1, B, n, n + 4]: the environment writes each step's observations straight into the next
        slice, see ``obs_slice``) and a state is a row number there (``st_row``).  Same HBM footprint as the worst-case
        compact copy, but recording a step moves only the acting agents' rows instead of re-reading and re-writing the
        whole observation tensor."""
        B, n, T = int(n_envs), int(n_nodes), int(horizon)
        w, A = n + 4, n + 1
        if capacity_steps is None:
            capacity_steps = B * T
        if capacity_entries is None:   # exact worst case for small batches, `entries_per_step` per env-step otherwise
            capacity_entries = B * T * n if B * T * n <= (1 << 22) else int(B * T * min(n, entries_per_step)) + B * n
        self.n_envs, self.n_nodes, self.horizon, self.device = B, n, T, device
        self.capacity_entries, self.capacity_steps = int(capacity_entries), int(capacity_steps)
        E, S = self.capacity_entries, self.capacity_steps
        f32, i32, u8 = torch.float32, torch.int32, torch.uint8
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)  # noqa: E731
        self.by_reference = bool(by_reference)
        self.t = OrderedDict(
            ent_obs=z((E, w), f32), ent_action=z((E, A), f32), ent_logp=z(E, f32), ent_env=z(E, i32),
            ent_agent=z(E, torch.int8), ent_adv=z(E, f32), ent_ret=z(E, f32),
            st_state=None if by_reference else z((S, n * w), f32), st_value=z(S, f32), st_env=z(S, i32), st_first=z(S, i32),
            st_k=z(S, u8),
            reward_tb=z((T, B), torch.float64), done_tb=z((T, B), u8), value_tb=z((T, B), f32), k_tb=z((T, B), u8),
            first_tb=z((T, B), i32), srow=z(B, i32), scan_scratch=z((B + 63) // 64, torch.int64), counters=z((T + 1, 4), torch.int64), err_flag=z(1, i32),
            obs_tb=z((T + 1, B, n, w), f32) if by_reference else None, st_row=z(S, i32) if by_reference else None)
        if by_reference and (T + 1) * B > 2**31 - 1:
            raise ValueError("the observation history would exceed 2^31 rows; use a smaller horizon or by_reference=False")
        self.struct = _lib.EmRollout(B, E, S, n, T, *[v.data_ptr() if v is not None else None for v in self.t.values()])
        # per-256-environment totals of acting agents, written by the environment kernel (em_step_out.chunk_totals) and
        # consumed by em_rollout_index_totals; two buffers used alternately (see MAPPO.record_and_act)
        self.chunk_totals = z((2, (B + 255) // 256), torch.int64)
        self.ref = C.byref(self.struct)
        self.steps_recorded = 0
        self.staged = RolloutBuffer()

    def clear(self):
        self.t["counters"].zero_()
        self.t["k_tb"].zero_()
        self.t["err_flag"].zero_()        # the overflow flag is sticky per batch, not per storage: the next batch starts clean
        self.steps_recorded = 0
        self.staged.clear()

    def obs_slice(self, i):
        """Slice ``i`` of the observation history (by-reference storage): where the observations at the START of recorded
        step ``i`` live.  Slices wrap, so a one-step scratch storage can be used ping-pong."""
        return self.t["obs_tb"][i % self.t["obs_tb"].shape[0]]

    def state_rows(self, n_states):
        """The first ``n_states`` state rows as a tensor [n_states, n*(n+4)] (copy), whatever the storage mode."""
        if not self.by_reference:
            return self.t["st_state"][:n_states]
        flat = self.t["obs_tb"].view(-1, self.n_nodes * (self.n_nodes + 4))
        return flat[self.t["st_row"][:n_states].long()]

    # the reference's list interface (mappo_algorithm.py:104-131)
    def add(self, obs, state, action, reward, done, log_prob, value):
        self.staged.add(obs, state, action, reward, done, log_prob, value)

    def get(self):
        return self.staged.get()

    @classmethod
    def from_lists(cls, data, n_nodes, device, value_norm=(0.0, 1.0)):
        """One-environment storage holding a flattened transition list (RolloutBuffer.get()): entry i is its own step
        (k = 1), so GAE chains through the list exactly as mappo_algorithm.py:250-258 and the critic is evaluated once per
        entry on that entry's own state, as mappo_algorithm.py:303 does."""
        import numpy as np
        L = len(data["rewards"])
        n = int(n_nodes)
        st = cls(1, n, L, device, capacity_entries=L, capacity_steps=L)
        f32 = torch.float32
        up = lambda a, dt: torch.as_tensor(np.asarray(a), dtype=dt).to(device)  # noqa: E731
        t = st.t
        t["ent_obs"].copy_(up(np.asarray(data["observations"], np.float32).reshape(L, n + 4), f32))
        t["st_state"].copy_(up(np.asarray(data["states"], np.float32).reshape(L, n * (n + 4)), f32))
        t["ent_action"].copy_(up(np.asarray(data["actions"], np.float32).reshape(L, n + 1), f32))
        t["ent_logp"].copy_(up(np.asarray(data["log_probs"], np.float32), f32))
        vals = up(np.asarray(data["values"], np.float32), f32)
        t["st_value"].copy_(vals)
        t["value_tb"].copy_(vals.view(L, 1))
        t["reward_tb"].copy_(up(np.asarray(data["rewards"], np.float64), torch.float64).view(L, 1))
        t["done_tb"].copy_(up(np.asarray(data["dones"], np.bool_).astype(np.uint8), torch.uint8).view(L, 1))
        t["k_tb"].fill_(1)
        t["st_k"].fill_(1)
        ar = torch.arange(L, dtype=torch.int32, device=device)
        t["first_tb"].copy_(ar.view(L, 1))
        t["st_first"].copy_(ar)
        c = torch.zeros((L + 1, 4), dtype=torch.int64)
        c[:, 0] = torch.arange(L + 1)
        c[:, 2] = torch.arange(L + 1)
        c[:L, 1] = 1
        c[:L, 3] = 1
        t["counters"].copy_(c.to(device))
        st.steps_recorded = L
        return st

    def totals_raw(self):
        """(entries, steps, error flag) recorded so far, without raising; synchronises."""
        c = self.t["counters"][self.steps_recorded].tolist()
        return int(c[0]), int(c[2]), int(self.t["err_flag"].item())

    def overflow_message(self, entries, steps):
        return (f"rollout storage capacity exceeded after {entries} entries / {steps} steps were recorded (capacity "
                f"{self.capacity_entries} entries, {self.capacity_steps} steps for {self.n_envs} envs x {self.horizon} "
                "steps): construct RolloutStorage / MAPPOTrainer with a larger capacity_entries")

    def totals(self):
        """(entries, steps) recorded so far; synchronises."""
        e, s, err = self.totals_raw()
        if err != 0:
            raise RuntimeError(self.overflow_message(e, s))
        return e, s

    def __len__(self):
        return (self.totals()[0] if self.steps_recorded else 0) + len(self.staged)


class MAPPO:
    """Same constructor as the reference (mappo_algorithm.py:145-160) plus ``n_envs``/``horizon`` for the
    device rollout storage.  Hyper-parameter attributes can be overridden after construction."""

    def __init__(self, obs_dim, state_dim, action_dim, n_agents, lr=3e-4, gamma=0.99, gae_lambda=0.95, clip_param=0.2,
                 value_loss_coeff=0.5, entropy_coeff=0.01, max_grad_norm=0.5, num_sgd_iter=4, device="cuda:0",
                 n_envs=1, horizon=1000, process_group=None, capacity_entries=None, entries_per_step=2.2, by_reference=True):
        self.obs_dim, self.state_dim, self.action_dim, self.n_agents = obs_dim, state_dim, action_dim, n_agents
        self.gamma, self.gae_lambda, self.clip_param = gamma, gae_lambda, clip_param
        self.value_loss_coeff, self.entropy_coeff = value_loss_coeff, entropy_coeff
        self.max_grad_norm, self.num_sgd_iter = max_grad_norm, num_sgd_iter
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("marl_amd.MAPPO runs on an MI355X (device='cuda:N'); there is no CPU path")
        self.process_group = process_group
        self.sync = GradSync(process_group)
        self._L = _lib.lib()
        # same construction order as the reference (actor then critic, on the CPU generator) so that
        # torch.manual_seed(seed) gives the reference's initial weights
        self.actor = MAPPOActor(obs_dim, action_dim)
        self.critic = MAPPOCritic(state_dim)
        n_a = sum(p.numel() for p in self.actor.parameters())
        n_c = sum(p.numel() for p in self.critic.parameters())
        self.flat_params = torch.zeros(n_a + n_c, dtype=torch.float32, device=self.device)
        assert self.actor.adopt(self.flat_params[:n_a]) == n_a and self.critic.adopt(self.flat_params[n_a:]) == n_c
        self.n_actor, self.n_critic = n_a, n_c
        self.flat_grads = torch.zeros(n_a + n_c + 8, dtype=torch.float32, device=self.device)   # + 2x3 loss sums
        self.actor_optimizer = HipAdam(self.actor, lr)
        self.critic_optimizer = HipAdam(self.critic, lr)
        self.value_normalizer = ValueNormalizer(1, device=self.device)
        self.buffer = RolloutStorage(n_envs, n_agents, horizon, self.device, capacity_entries=capacity_entries,
                                     entries_per_step=entries_per_step, by_reference=by_reference)
        self.moments = torch.zeros(3, dtype=torch.float64, device=self.device)
        self._gae_partials = torch.zeros(3 * ((n_envs + 255) // 256), dtype=torch.float64, device=self.device)
        self._slab_a = self._slab_c = None
        self.grad_norms = torch.zeros(2, dtype=torch.float32, device=self.device)
        self.last_update = {}
        # nothing before GAE reads the critic's values: during a recorded rollout the trainer lets the value forward run on
        # a side stream beside the policy forward and the environment step (overlap_value); join_value() orders the main
        # stream behind it again (end of the episode, compute_gae)
        self.overlap_value = False
        self._side = self._ev_idx = self._ev_val = None
        self._value_pending = False
        # With the states held by reference the critic does not run during the rollout at all: one launch over every
        # recorded state before GAE (em_value_all) replaces a thousand small ones; same rows, same kernel, same bits.
        self.defer_values = True
        # index/compaction and the actor forward of a step in one launch (em_rollout_policy_step) where the deferred mode applies
        self.fuse_index_policy = True
        self._values_through = 0          # recorded steps whose values have been computed

    def join_value(self):
        if self._value_pending:
            torch.cuda.current_stream(self.device).wait_event(self._ev_val)
            self._value_pending = False

    def _deferred(self):
        return self.defer_values and self.buffer.by_reference and self.buffer.horizon > 1

    def compute_values(self):
        """Values of every recorded state that does not have one yet (deferred mode); a no-op otherwise."""
        ro = self.buffer
        if not self._deferred() or ro.steps_recorded == 0 or self._values_through == ro.steps_recorded:
            return
        n_states = int(ro.t["counters"][ro.steps_recorded, 2].item())
        mean, var = self._norm()
        c_net = self.critic.net()
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_value_all(ro.ref, n_states, C.byref(c_net), mean, var, _stream(self.device)), "em_value_all")
        self._values_through = ro.steps_recorded

    # -- rollout side ----------------------------------------------------------------------------------
    def _norm(self):
        return float(self.value_normalizer.running_mean.reshape(-1)[0]), float(self.value_normalizer.running_var.reshape(-1)[0])

    def record_and_act(self, t, obs, active, rank, placement_out, resource_out, explore=True, noise=None, seed=0,
                       episode=0, env_offset=0, rng_step=None, totals=None, totals_next=None):
        """Steps 199-219 of MAPPOTrainer.train_episode for all environments: index the acting agents, sample
        their actions, format the env actions and evaluate the critic on the pre-step state."""
        ro = self.buffer
        a_ref, c_ref, log_std = self._nets_cached          # structs of begin_rollout: parameters live in one flat buffer
        mean, var = self._norm_cached
        L = self._L
        with torch.cuda.device(self.device):
            main = torch.cuda.current_stream(self.device)
            s = int(main.cuda_stream)
            if (totals is not None and self.fuse_index_policy and self._deferred() and 2 <= self.n_agents <= 12):
                # one launch: entries numbered and copied, actor evaluated on them (bit-identical to the two calls below)
                _lib.check(L.em_rollout_policy_step(
                    ro.ref, t, active.data_ptr(), rank.data_ptr(), obs.data_ptr(), totals.data_ptr(),
                    totals_next.data_ptr() if totals_next is not None else None, a_ref, log_std,
                    noise.data_ptr() if noise is not None else None, 1 if explore else 0, int(seed) & (2**64 - 1), int(episode),
                    int(t if rng_step is None else rng_step), int(env_offset), placement_out.data_ptr(), resource_out.data_ptr(),
                    s), "em_rollout_policy_step")
                ro.steps_recorded = t + 1
                return
            if totals is not None:       # the environment kernel already counted the acting agents per chunk
                _lib.check(L.em_rollout_index_totals(ro.ref, t, active.data_ptr(), rank.data_ptr(), obs.data_ptr(),
                                                     totals.data_ptr(), totals_next.data_ptr() if totals_next is not None else None,
                                                     s), "em_rollout_index_totals")
            else:
                _lib.check(L.em_rollout_index(ro.ref, t, active.data_ptr(), rank.data_ptr(), obs.data_ptr(), s), "em_rollout_index")
            fused_critic = c_ref
            if self._deferred():
                fused_critic = None
            elif self.overlap_value:
                if self._side is None:
                    self._side = torch.cuda.Stream(self.device)
                    self._ev_idx, self._ev_val = torch.cuda.Event(), torch.cuda.Event()
                self._ev_idx.record(main)                   # this step's index kernels
                self._side.wait_event(self._ev_idx)
                _lib.check(L.em_value_step(ro.ref, t, c_ref, mean, var, int(self._side.cuda_stream)), "em_value_step")
                self._ev_val.record(self._side)
                self._value_pending = True
                fused_critic = None
            _lib.check(L.em_policy_step(ro.ref, t, a_ref, log_std, fused_critic, mean, var,
                                        noise.data_ptr() if noise is not None else None, 1 if explore else 0,
                                        int(seed) & (2**64 - 1), int(episode), int(t if rng_step is None else rng_step),
                                        int(env_offset), placement_out.data_ptr(), resource_out.data_ptr(), s), "em_policy_step")
        ro.steps_recorded = t + 1

    def begin_rollout(self):
        self.buffer.clear()
        self._values_through = 0
        self._norm_cached = self._norm()
        self._cache_nets()

    def _cache_nets(self):
        self._net_structs = (self.actor.net(), self.critic.net())          # keeps the ctypes structs alive
        self._nets_cached = (C.byref(self._net_structs[0]), C.byref(self._net_structs[1]), self.actor.log_std.data_ptr())

    # -- GAE (mappo_algorithm.py:241-262) --------------------------------------------------------------
    def compute_gae(self, rewards=None, values=None, dones=None, next_value=0.0):
        """Without arguments: GAE over the device rollout storage (the batched trainer's path), results in the storage.
        With the reference's arguments (mappo_algorithm.py:241: flattened lists + ``next_value``): the same ``em_gae``
        kernel on a one-environment upload of the lists, returns ``(advantages, returns)`` as python lists.  The
        recursion runs in float64 like the reference's; the kernel stores float32, which is the cast the reference
        applies to both lists right afterwards (mappo_algorithm.py:289-290)."""
        if rewards is not None:
            return self._compute_gae_lists(rewards, values, dones, next_value)
        ro = self.buffer
        self.join_value()
        self.compute_values()
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_gae(C.byref(ro.struct), float(self.gamma), float(self.gae_lambda),
                                      self._gae_partials.data_ptr(), self._gae_partials.numel(), self.moments.data_ptr(),
                                      _stream(self.device)), "em_gae")

    def _compute_gae_lists(self, rewards, values, dones, next_value):
        import numpy as np
        L = len(rewards)
        if L == 0:
            return [], []
        r = np.asarray(rewards, np.float64).copy()
        d = np.asarray(dones, np.bool_)
        # next_value closes the list: delta_last = r + gamma * next_value * (1 - done) - v (mappo_algorithm.py:251-256);
        # em_gae closes with 0, so the term is folded into the last reward (same float64 sum, same order)
        r[-1] = r[-1] + float(self.gamma) * float(next_value) * (1.0 - float(d[-1]))
        data = {"observations": np.zeros((L, self.obs_dim), np.float32), "states": np.zeros((L, self.state_dim), np.float32),
                "actions": np.zeros((L, self.action_dim), np.float32), "log_probs": np.zeros(L, np.float32),
                "rewards": r, "dones": d, "values": np.asarray(values, np.float64)}
        st = RolloutStorage.from_lists(data, self.n_agents, self.device)
        part = torch.zeros(3, dtype=torch.float64, device=self.device)
        mom = torch.zeros(3, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_gae(st.ref, float(self.gamma), float(self.gae_lambda), part.data_ptr(), 3, mom.data_ptr(),
                                      _stream(self.device)), "em_gae")
        return st.t["ent_adv"][:L].double().cpu().tolist(), st.t["ent_ret"][:L].double().cpu().tolist()

    # -- update (mappo_algorithm.py:264-345) -----------------------------------------------------------
    def _waves(self, rows, per_cu=1):
        tiles = (rows + 15) // 16
        blocks = max(1, min(256 * per_cu, (tiles + 3) // 4))
        return 4 * blocks

    def update(self):
        ro = self.buffer
        if len(ro.staged) and ro.steps_recorded:
            # two sources at once: the list interface (buffer.add) and a device-recorded episode.  The reference has ONE
            # list; merging a host list into env-major device records has no defined order, and dropping either silently
            # would lose transitions
            raise RuntimeError(f"{len(ro.staged)} transitions were added through buffer.add() while {ro.steps_recorded} recorded "
                               "steps of a batched episode are pending: call update() after each kind, not after both")
        if len(ro.staged) and ro.steps_recorded == 0:
            # transitions added through the reference's list interface (buffer.add, mappo_trainer.py:160-175)
            listed = RolloutStorage.from_lists(ro.staged.get(), self.n_agents, self.device)
            ro.staged.clear()
            self.buffer = listed
            try:
                return self.update()
            finally:
                self.buffer = ro
        # Every decision that would leave before a collective is taken on GLOBAL totals, so all ranks return, raise or
        # go on together: a rank whose shard recorded nothing still joins every all-reduce (with zero gradients), and an
        # overflow on one rank raises on all of them instead of leaving the others waiting in all_reduce.
        E, S, err = ro.totals_raw() if ro.steps_recorded else (0, 0, 0)
        recorded_g, E_g, err_g = self.sync.sum_counts([ro.steps_recorded, E, err], self.device)
        if err_g:
            msg = ro.overflow_message(E, S) if err else "rollout storage capacity exceeded on another rank"
            ro.clear()
            self._values_through = 0
            raise RuntimeError(msg)
        if recorded_g == 0 or E_g == 0:
            ro.clear()
            self._values_through = 0
            return {}
        if ro.steps_recorded:
            self.compute_gae()
        else:
            self.moments.zero_()
        n_global = self.sync.sum_moments(self.moments)   # global advantage moments and transition count
        L = self._L
        na, nc = self.n_actor, self.n_critic
        # the 14-wide actor kernel is built for two workgroups per CU (mlp3_bwd_kernel, KU == 1)
        wa, wc = self._waves(E, getattr(self, "actor_wg_per_cu", 2 if self.obs_dim <= 16 else 1)), self._waves(S)
        sa = int(L.em_ppo_slab_floats(self.obs_dim, self.action_dim, 1))
        sc = int(L.em_ppo_slab_floats(self.state_dim, 1, 0))
        if self._slab_a is None or self._slab_a.numel() < wa * sa:
            self._slab_a = torch.empty(wa * sa, dtype=torch.float32, device=self.device)
        if self._slab_c is None or self._slab_c.numel() < wc * sc:
            self._slab_c = torch.empty(wc * sc, dtype=torch.float32, device=self.device)
        mean, var = self._norm()
        g = self.flat_grads
        loss_a, loss_c = g[na + nc:na + nc + 3], g[na + nc + 4:na + nc + 7]
        t = ro.t
        acc = torch.zeros(8, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            s = _stream(self.device)
            for _ in range(int(self.num_sgd_iter)):
                a_net, c_net = self.actor.net(), self.critic.net()
                if E == 0:             # empty shard: contributes zero gradients and loss sums to the all-reduce
                    g.zero_()
                else:
                    self._local_grads(L, a_net, c_net, t, E, S, n_global, mean, var, wa, wc, g, loss_a, loss_c, s)
                self.sync.sum_grads(g)   # the one data-path collective per epoch: gradient-of-sums + loss sums
                self.actor_optimizer.step_count += 1
                self.critic_optimizer.step_count += 1
                ga, gc = self.actor_optimizer.param_groups[0], self.critic_optimizer.param_groups[0]
                _lib.check(L.em_adam_step(self.actor.flat.data_ptr(), g.data_ptr(), self.actor_optimizer.exp_avg.data_ptr(),
                                          self.actor_optimizer.exp_avg_sq.data_ptr(), na, float(ga["lr"]),
                                          self.critic.flat.data_ptr(), g[na:].data_ptr(),
                                          self.critic_optimizer.exp_avg.data_ptr(), self.critic_optimizer.exp_avg_sq.data_ptr(),
                                          nc, float(gc["lr"]), self.actor_optimizer.step_count, float(ga["betas"][0]),
                                          float(ga["betas"][1]), float(ga["eps"]), float(self.max_grad_norm),
                                          self.grad_norms.data_ptr(), s), "em_adam_step")
                acc += g[na + nc:]          # per-epoch loss sums (already global after the all-reduce)
        ro.clear()
        self._values_through = 0
        sums = acc.cpu()
        _require_finite(sums, self.grad_norms)
        k = float(self.num_sgd_iter) * n_global
        self.last_update = {"actor_loss": float(-sums[0] / k), "critic_loss": float(0.5 * sums[4] / k),
                            "entropy": float(sums[1] / k), "transitions": n_global}
        return {"actor_loss": self.last_update["actor_loss"], "critic_loss": self.last_update["critic_loss"],
                "entropy": self.last_update["entropy"]}

    def _local_grads(self, L, a_net, c_net, t, E, S, n_global, mean, var, wa, wc, g, loss_a, loss_c, s):
        """This rank's gradient-of-sums of one epoch into ``g`` (actor | critic | loss sums)."""
        na = self.n_actor
        _lib.check(L.em_ppo_actor_grad(C.byref(a_net), self.actor.log_std.data_ptr(), t["ent_obs"].data_ptr(),
                                       t["ent_action"].data_ptr(), t["ent_logp"].data_ptr(), t["ent_adv"].data_ptr(),
                                       self.moments.data_ptr(), E, n_global, float(self.clip_param),
                                       float(self.entropy_coeff), self._slab_a.data_ptr(), wa, g.data_ptr(),
                                       loss_a.data_ptr(), s), "em_ppo_actor_grad")
        by_ref = t["st_state"] is None
        _lib.check(L.em_ppo_critic_grad_rows(C.byref(c_net), (t["obs_tb"] if by_ref else t["st_state"]).data_ptr(),
                                             t["st_row"].data_ptr() if by_ref else None, t["st_first"].data_ptr(),
                                             t["st_k"].data_ptr(), t["ent_ret"].data_ptr(), S, n_global,
                                             float(self.value_loss_coeff), mean, var, self._slab_c.data_ptr(), wc,
                                             g[na:].data_ptr(), loss_c.data_ptr(), s), "em_ppo_critic_grad")

    # -- dict API of the reference (mappo_algorithm.py:189-239), one environment -------------------------------
    def get_actions(self, observations, explore=True, noise=None):
        """``{agent_id: obs}`` -> ``({agent_id: action float32[action_dim]}, {agent_id: log_prob float})``
        (mappo_algorithm.py:189-224).  explore: a = mu + sigma * eps with eps drawn per agent, in dict order, as
        ``Normal.rsample()`` draws it on the CPU generator (or taken from ``noise`` [agents, action_dim]); eval: a = mu and
        log-prob 0.0.  The arithmetic runs in the HIP kernels (em_mlp3_forward, em_gauss_sample, em_gauss_logp)."""
        import numpy as np
        ids = list(observations.keys())
        if not ids:
            return {}, {}
        x = torch.as_tensor(np.stack([np.asarray(observations[a], np.float32).reshape(-1) for a in ids])).to(self.device)
        if explore:
            if noise is None:
                A = self.action_dim
                eps = torch.cat([torch.normal(torch.zeros(1, A), torch.ones(1, A)) for _ in ids])
            else:
                eps = torch.as_tensor(np.asarray(noise, np.float32)).reshape(len(ids), self.action_dim)
            mu = self.actor(x)
            act = torch.empty_like(mu)
            with torch.cuda.device(self.device):
                _lib.check(self._L.em_gauss_sample(mu.data_ptr(), self.actor.log_std.data_ptr(), eps.to(self.device).data_ptr(),
                                                   len(ids), self.action_dim, act.data_ptr(), _stream(self.device)),
                           "em_gauss_sample")
                logp = torch.empty(len(ids), dtype=torch.float32, device=self.device)
                _lib.check(self._L.em_gauss_logp(mu.data_ptr(), self.actor.log_std.data_ptr(), act.data_ptr(), len(ids),
                                                 self.action_dim, logp.data_ptr(), None, _stream(self.device)), "em_gauss_logp")
            act, logp = act.cpu().numpy(), logp.cpu().numpy()
            return {a: act[i] for i, a in enumerate(ids)}, {a: float(logp[i]) for i, a in enumerate(ids)}
        mu = self.actor(x).cpu().numpy()
        return {a: mu[i] for i, a in enumerate(ids)}, {a: 0.0 for a in ids}

    def get_values(self, state):
        x = torch.as_tensor(state, dtype=torch.float32, device=self.device).reshape(1, -1)
        mean, var = self._norm()
        return float(self.critic(x)[0, 0]) * math.sqrt(var + 1e-8) + mean

    # -- checkpoint (mappo_algorithm.py:347-364): same keys and tensor names -----------------------------------
    def save(self, path):
        torch.save({"actor": {k: v.detach().cpu().clone() for k, v in self.actor.state_dict().items()},
                    "critic": {k: v.detach().cpu().clone() for k, v in self.critic.state_dict().items()},
                    "actor_optimizer": _cpu(self.actor_optimizer.state_dict()),
                    "critic_optimizer": _cpu(self.critic_optimizer.state_dict()),
                    "value_normalizer": {k: v.detach().cpu().clone() for k, v in self.value_normalizer.state_dict().items()}},
                   path)

    def load(self, path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        with torch.no_grad():
            for mod, key in ((self.actor, "actor"), (self.critic, "critic")):
                own = dict(mod.named_parameters())
                for name, val in ck[key].items():
                    own[name].copy_(val.to(self.device))
        self.actor_optimizer.load_state_dict(ck["actor_optimizer"])
        self.critic_optimizer.load_state_dict(ck["critic_optimizer"])
        self.value_normalizer.load_state_dict({k: v.to(self.device) for k, v in ck["value_normalizer"].items()})


def _require_finite(loss_sums, grad_norms=None):
    """The update's only silent failure mode made loud: a non-finite loss sum or gradient norm.  The Adam kernels skip a
    step whose gradient norm is not finite (parameters stay as they were), and this raises.  Causes: a diverged run, or a
    lost hand-over between the two wavefronts of the split backward kernels (their bounded wait poisons the slab with NaN;
    EM_BWD_SPLIT=0 selects the one-wavefront kernels, bitwise the same results)."""
    bad = not bool(torch.isfinite(loss_sums).all())
    if grad_norms is not None:
        bad = bad or not bool(torch.isfinite(grad_norms).all().item())
    if bad:
        raise RuntimeError("update(): non-finite loss sums or gradient norm; the optimiser step of the affected network was "
                           "skipped (parameters unchanged).  If the inputs are finite, set EM_BWD_SPLIT=0 (one-wavefront "
                           "backward kernels) and report the case")


def _cpu(obj):
    if isinstance(obj, torch.Tensor):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_cpu(v) for v in obj)
    return obj
