"""EdgeSimGym on the scaled topology (BASELINE.json configs[4]): D decider agents over S edge servers, batched.

The reference only has deciders == servers (MRRL.PY:150-162,355); ``include/edgemarl.h`` states the generalisation
(home server d mod S) and ``csrc/edge_env_scaled.hip`` implements it, one wavefront per environment.  With
``n_deciders == n_nodes`` it is the reference step bit for bit.  Outputs are COMPACT: the pending list (deciders that
hold a task, in arrival order = the reference's obs-dict insertion order) and one observation row per pending decider,
instead of a dense ``[D, S+4]`` tensor that would be >99 % zeros."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import SCALED_ENV_CONFIG, make_em_config
from .config import latency_log_capacity as default_latency_log_capacity
from .env import _stream_ptr

PENDING_CAP = 64


class ScaledEdgeSimGym:
    """reset(seed=None) -> (n_pending[B] i32, pend_decider[B,64] i32, pend_obs[B,64,S+4] f32)
    step(placement, resource, by_slot=False) -> (n_pending, pend_decider, pend_obs, reward[B] f64, done[B] u8, info)

    ``placement``/``resource`` are ``[B, D]`` indexed by decider (the dict form of MRRL.PY:280-281), or ``[B, 64]``
    aligned with the pending list of the previous call when ``by_slot`` is set.  Only the first ``n_pending[b]`` rows of
    the list outputs are meaningful.  ``reward`` applies the consumer rule of mappo_trainer.py:209-212."""

    def __init__(self, config=None, n_envs: int = 1, device="cuda:0", seed: int = 0, env_offset: int = 0,
                 latency_log_capacity: int | None = None):
        self.config = dict(SCALED_ENV_CONFIG if config is None else config)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ScaledEdgeSimGym runs on an MI355X (device='cuda:N'); there is no CPU path")
        self.n_envs = int(n_envs)
        self.n_nodes = int(self.config["n_nodes"])
        self.n_deciders = int(self.config.get("n_deciders", self.n_nodes))
        self.episode_len = int(self.config["episode_length"])
        if latency_log_capacity is None:   # exact p99 needs every latency of the episode (MRRL.PY:476-479)
            latency_log_capacity = default_latency_log_capacity(self.config, self.n_envs, min(self.n_deciders, PENDING_CAP))
        self.latency_log_capacity = int(latency_log_capacity)
        self._cfg = make_em_config(self.config)
        self._L = _lib.lib()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._h = C.c_void_p()
        with torch.cuda.device(dev_index):
            _lib.check(self._L.ems_create(C.byref(self._cfg), self.n_deciders, self.n_envs, dev_index,
                                          int(seed) & (2**64 - 1), int(env_offset), self.latency_log_capacity,
                                          C.byref(self._h)), "ems_create")
        B, S, d = self.n_envs, self.n_nodes, self.device
        self.n_pending = torch.zeros(B, dtype=torch.int32, device=d)
        self.pend_decider = torch.zeros((B, PENDING_CAP), dtype=torch.int32, device=d)
        self.pend_obs = torch.zeros((B, PENDING_CAP, S + 4), dtype=torch.float32, device=d)
        self.reward_raw = torch.zeros(B, dtype=torch.float64, device=d)
        self.reward = torch.zeros(B, dtype=torch.float64, device=d)
        self.emitted = torch.zeros(B, dtype=torch.uint8, device=d)
        self.done = torch.zeros(B, dtype=torch.uint8, device=d)
        self.step_latency = None
        self._out = _lib.EmsStepOut()
        self._bind()
        self._disable_env_exploration = bool(self.config.get("disable_env_exploration", False))

    def _bind(self):
        o = self._out
        o.n_pending, o.pend_decider, o.pend_obs = self.n_pending.data_ptr(), self.pend_decider.data_ptr(), self.pend_obs.data_ptr()
        o.reward, o.reward_seen = self.reward_raw.data_ptr(), self.reward.data_ptr()
        o.emitted, o.done = self.emitted.data_ptr(), self.done.data_ptr()
        o.step_latency = self.step_latency.data_ptr() if self.step_latency is not None else None

    def enable_debug_outputs(self):
        self.step_latency = torch.zeros((self.n_envs, PENDING_CAP), dtype=torch.float64, device=self.device)
        self._bind()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.ems_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def disable_env_exploration(self) -> bool:
        return self._disable_env_exploration

    @disable_env_exploration.setter
    def disable_env_exploration(self, v: bool):
        self._disable_env_exploration = bool(v)
        _lib.check(self._L.ems_set_exploration(self._h, int(bool(v)), self.global_step_counter))

    @property
    def global_step_counter(self) -> int:
        return int(self._L.ems_get_exploration_counter(self._h))

    @global_step_counter.setter
    def global_step_counter(self, v: int):
        _lib.check(self._L.ems_set_exploration(self._h, int(self._disable_env_exploration), int(v)))

    def set_envs_per_wave(self, envs_per_wave: int):
        """Kernel variant (4 environments per wavefront by default, or 1); tuning/tests only, identical results."""
        _lib.check(self._L.ems_set_envs_per_wave(self._h, int(envs_per_wave)))

    @property
    def record_bytes(self) -> int:
        return int(self._L.ems_record_bytes(self._h))

    def reset(self, *, seed=None, options=None):
        with torch.cuda.device(self.device):
            _lib.check(self._L.ems_reset(self._h, 0 if seed is None else 1, 0 if seed is None else int(seed) & (2**64 - 1),
                                         C.byref(self._out), _stream_ptr(self.device)), "ems_reset")
        return self.n_pending, self.pend_decider, self.pend_obs

    def step(self, placement: torch.Tensor, resource: torch.Tensor, by_slot: bool = False,
             reward_seen_out: torch.Tensor | None = None, done_out: torch.Tensor | None = None):
        """``reward_seen_out`` / ``done_out`` ([B] float64 / uint8, contiguous) redirect those two outputs of this call,
        e.g. straight into a row of the rollout storage."""
        B = self.n_envs
        width = PENDING_CAP if by_slot else self.n_deciders
        if placement.dtype != torch.int32 or resource.dtype != torch.float32:
            raise TypeError("placement must be int32 and resource float32")
        if tuple(placement.shape) != (B, width) or tuple(resource.shape) != (B, width):
            raise ValueError(f"placement/resource must have shape [{B}, {width}]")
        if placement.device != self.device or resource.device != self.device:
            raise ValueError("actions must live on the environment's device")
        if not (placement.is_contiguous() and resource.is_contiguous()):
            raise ValueError("actions must be contiguous")
        o = self._out
        if reward_seen_out is not None:
            o.reward_seen = reward_seen_out.data_ptr()
        if done_out is not None:
            o.done = done_out.data_ptr()
        try:
            with torch.cuda.device(self.device):
                _lib.check(self._L.ems_step(self._h, placement.data_ptr(), resource.data_ptr(), int(bool(by_slot)),
                                            C.byref(o), _stream_ptr(self.device)), "ems_step")
        finally:
            o.reward_seen, o.done = self.reward.data_ptr(), self.done.data_ptr()
        info = {"reward_raw": self.reward_raw, "emitted": self.emitted}
        return self.n_pending, self.pend_decider, self.pend_obs, self.reward, self.done, info

    def episode_metrics(self) -> torch.Tensor:
        out = torch.zeros((self.n_envs, 6), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.ems_episode_metrics(self._h, out.data_ptr(), _stream_ptr(self.device)), "ems_episode_metrics")
        return out

    def get_state(self) -> dict:
        B, S = self.n_envs, self.n_nodes
        st = {"load": np.zeros((B, S)), "queue": np.zeros((B, S)), "sim_time": np.zeros(B), "elapsed": np.zeros(B),
              "step": np.zeros(B, np.int32), "counts": np.zeros((B, 3), np.int64)}
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.ems_get_state(self._h, p(st["load"]), p(st["queue"]), p(st["sim_time"]), p(st["elapsed"]),
                                             p(st["step"]), p(st["counts"]), _stream_ptr(self.device)), "ems_get_state")
        return st

    def dense_obs(self) -> torch.Tensor:
        """Dense ``[B, D, S+4]`` view of the compact observation list (zero rows for idle deciders): the layout of
        ``MAPPOTrainer.obs_dict_to_state`` (mappo_trainer.py:101-123).  Small batches / tests only."""
        B, D, S = self.n_envs, self.n_deciders, self.n_nodes
        dense = torch.zeros((B, D, S + 4), dtype=torch.float32, device=self.device)
        slot = torch.arange(PENDING_CAP, device=self.device)[None, :]
        live = slot < self.n_pending[:, None]
        b_idx = torch.arange(B, device=self.device)[:, None].expand(B, PENDING_CAP)[live]
        dense[b_idx, self.pend_decider[live].long()] = self.pend_obs[live]
        return dense
