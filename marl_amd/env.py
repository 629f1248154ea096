"""Batched EdgeSimGym on MI355X: the reference's ``reset()/step(actions)`` API (MRRL.PY:188,386) with a
leading batch-of-environments dimension.  All arithmetic happens in the HIP kernels of
``csrc/edge_env.hip``; torch only owns the device buffers and the stream."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import ENV_CONFIG, make_em_config
from .config import latency_log_capacity as default_latency_log_capacity

METRIC_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "violation_rate", "throughput_tps",
               "load_balance_jain")

_TAPE_F64 = ("explore_u", "explore_res", "bw_u", "net_lat", "rate_u", "dt_raw", "task_workload", "task_data",
             "task_deadline_offset")
_TAPE_I32 = ("explore_tgt", "n_arrivals", "task_decider")


def _stream_ptr(device) -> int:
    return int(torch.cuda.current_stream(device).cuda_stream)


class DeviceTape:
    """One step of pre-drawn random numbers on the device (``em_tape``).  ``arrays`` maps the field
    names of ``em_tape`` to array-likes with leading dimension n_envs; missing optional fields stay NULL."""

    def __init__(self, arrays: dict, device):
        self.tensors = {}
        st = _lib.EmTape()
        for k, v in arrays.items():
            if k in _TAPE_F64:
                t = torch.as_tensor(np.ascontiguousarray(v, np.float64)).to(device)
            elif k in _TAPE_I32:
                t = torch.as_tensor(np.ascontiguousarray(v, np.int32)).to(device)
            else:
                raise KeyError(f"unknown tape field {k}")
            self.tensors[k] = t
            setattr(st, k, t.data_ptr())
        st.arrival_slots = int(self.tensors["task_workload"].shape[-1]) if "task_workload" in self.tensors else 0
        self.struct = st


class BatchedEdgeSimGym:
    """``n_envs`` independent EdgeSimGym instances stepped by one kernel launch.

    reset(seed=None) -> (obs[B,N,N+4] f32, active[B] int64 bitmask)
    step(placement[B,N] i32, resource[B,N] f32) -> (obs, reward[B] f64, done[B] u8, truncated[B] u8, info)

    ``reward`` already applies the reference's consumer rule (mappo_trainer.py:209-212): it is 0 when
    the new observation set is empty; ``info["reward_raw"]``/``info["emitted"]`` hold the two parts.
    Returned tensors are views of buffers owned by the environment and are overwritten by the next call.
    """

    def __init__(self, config=None, n_envs: int = 1, device="cuda:0", seed: int = 0, env_offset: int = 0,
                 latency_log_capacity: int | None = None):
        self.config = dict(ENV_CONFIG if config is None else config)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchedEdgeSimGym runs on an MI355X (device='cuda:N'); there is no CPU path")
        self.n_envs = int(n_envs)
        self.n_nodes = int(self.config["n_nodes"])
        self.episode_len = int(self.config["episode_length"])
        self.cpu_caps = np.asarray(self.config["node_cpu_capacity"], np.float64)
        self._cfg = make_em_config(self.config)
        if latency_log_capacity is None:   # exact p99 needs every latency of the episode (MRRL.PY:476-479)
            latency_log_capacity = default_latency_log_capacity(self.config, self.n_envs, self.n_nodes)
        self.latency_log_capacity = int(latency_log_capacity)
        self._L = _lib.lib()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._h = C.c_void_p()
        with torch.cuda.device(dev_index):
            _lib.check(self._L.em_create(C.byref(self._cfg), self.n_envs, dev_index, int(seed) & (2**64 - 1),
                                         int(env_offset), self.latency_log_capacity, C.byref(self._h)), "em_create")
        B, n = self.n_envs, self.n_nodes
        d = self.device
        self.obs = torch.zeros((B, n, n + 4), dtype=torch.float32, device=d)
        self.active = torch.zeros(B, dtype=torch.int64, device=d)
        self.rank = torch.full((B, n), -1, dtype=torch.int8, device=d)
        self.reward_raw = torch.zeros(B, dtype=torch.float64, device=d)
        self.reward = torch.zeros(B, dtype=torch.float64, device=d)
        self.emitted = torch.zeros(B, dtype=torch.uint8, device=d)
        self.done = torch.zeros(B, dtype=torch.uint8, device=d)
        self.truncated = torch.zeros(B, dtype=torch.uint8, device=d)
        self.step_latency = None
        self.n_draws = None
        self._out = _lib.EmStepOut()
        self._bind_outputs()
        self._disable_env_exploration = bool(self.config.get("disable_env_exploration", False))

    # ------------------------------------------------------------------------------------------
    def _bind_outputs(self):
        o = self._out
        o.obs, o.active, o.rank = self.obs.data_ptr(), self.active.data_ptr(), self.rank.data_ptr()
        o.reward, o.emitted, o.done = self.reward_raw.data_ptr(), self.emitted.data_ptr(), self.done.data_ptr()
        o.step_latency = self.step_latency.data_ptr() if self.step_latency is not None else None
        o.n_draws = self.n_draws.data_ptr() if self.n_draws is not None else None
        o.reward_seen = self.reward.data_ptr()

    def bind_obs(self, obs: torch.Tensor | None = None, chunk_totals: torch.Tensor | None = None):
        """Redirect the observation output of the following reset/step calls into ``obs`` ([B, N, N+4] float32,
        contiguous, e.g. a slice of a rollout's observation history) - ``self.obs`` then IS that tensor - and/or let the
        kernel add its per-256-environment acting-agent totals into ``chunk_totals`` (``em_step_out.chunk_totals``;
        ``None`` switches that output off)."""
        if obs is not None:
            B, n = self.n_envs, self.n_nodes
            if tuple(obs.shape) != (B, n, n + 4) or obs.dtype != torch.float32 or not obs.is_contiguous() or obs.device != self.device:
                raise ValueError(f"obs must be a contiguous float32 [{B}, {n}, {n + 4}] tensor on {self.device}")
            self.obs = obs
            self._out.obs = obs.data_ptr()
        self._out.chunk_totals = chunk_totals.data_ptr() if chunk_totals is not None else None

    def enable_debug_outputs(self):
        """Also produce per-slot latencies and draw counts each step (parity tests)."""
        B, n = self.n_envs, self.n_nodes
        self.step_latency = torch.zeros((B, n), dtype=torch.float64, device=self.device)
        self.n_draws = torch.zeros(B, dtype=torch.int32, device=self.device)
        self._bind_outputs()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.em_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- exploration switch/counter (MRRL.PY:122,179; mappo_trainer.py:336-346) ------------------
    @property
    def disable_env_exploration(self) -> bool:
        return self._disable_env_exploration

    @disable_env_exploration.setter
    def disable_env_exploration(self, v: bool):
        self._disable_env_exploration = bool(v)
        _lib.check(self._L.em_set_exploration(self._h, int(bool(v)), self.global_step_counter))

    @property
    def global_step_counter(self) -> int:
        return int(self._L.em_get_exploration_counter(self._h))

    @global_step_counter.setter
    def global_step_counter(self, v: int):
        _lib.check(self._L.em_set_exploration(self._h, int(self._disable_env_exploration), int(v)))

    def set_lanes_per_env(self, lanes: int):
        _lib.check(self._L.em_set_lanes_per_env(self._h, int(lanes)))

    def set_workgroup_waves(self, waves: int):
        """Wavefronts per workgroup of the step kernel (waves - 1 workers + 1 scalar); tuning only."""
        _lib.check(self._L.em_set_workgroup_waves(self._h, int(waves)))

    def kernel_variant(self) -> dict:
        """Layout the next launch uses (``em_get_kernel_variant``): benchmark/profile labels."""
        lanes, waves = C.c_int32(0), C.c_int32(0)
        packed = self._L.em_get_kernel_variant(self._h, C.byref(lanes), C.byref(waves))
        if packed < 0:
            _lib.check(packed, "em_get_kernel_variant")
        return {"lanes_per_env": int(lanes.value), "packed": bool(packed), "workgroup_waves": int(waves.value),
                "kernel": f"env_kernel<{'packed' if packed else lanes.value},philox,step>"}

    @property
    def record_bytes(self) -> int:
        return int(self._L.em_record_bytes(self._h))

    # -- API ---------------------------------------------------------------------------------------
    def reset(self, *, seed=None, options=None, tape: DeviceTape | None = None):
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_reset(self._h, C.byref(tape.struct) if tape is not None else None,
                                        0 if seed is None else 1, 0 if seed is None else int(seed) & (2**64 - 1),
                                        C.byref(self._out), _stream_ptr(self.device)), "em_reset")
        return self.obs, self.active

    def step(self, placement: torch.Tensor, resource: torch.Tensor, tape: DeviceTape | None = None,
             reward_seen_out: torch.Tensor | None = None, done_out: torch.Tensor | None = None):
        """``reward_seen_out`` / ``done_out`` ([B] float64 / uint8, contiguous) redirect those two outputs of this
        call, e.g. straight into a row of the rollout storage."""
        B, n = self.n_envs, self.n_nodes
        if placement.dtype != torch.int32 or resource.dtype != torch.float32:
            raise TypeError("placement must be int32 and resource float32")
        if placement.shape != (B, n) or resource.shape[:2] != (B, n) or resource.numel() != B * n:
            raise ValueError(f"placement/resource must have shape [{B}, {n}]")
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
                _lib.check(self._L.em_step(self._h, placement.data_ptr(), resource.data_ptr(),
                                           C.byref(tape.struct) if tape is not None else None, C.byref(o),
                                           _stream_ptr(self.device)), "em_step")
        finally:
            o.reward_seen, o.done = self.reward.data_ptr(), self.done.data_ptr()
        info = {"reward_raw": self.reward_raw, "emitted": self.emitted, "rank": self.rank, "active": self.active}
        return self.obs, self.reward, self.done, self.truncated, info

    def _step_result(self):
        info = {"reward_raw": self.reward_raw, "emitted": self.emitted, "rank": self.rank, "active": self.active}
        return self.obs, self.reward, self.done, self.truncated, info

    def step_raw(self, raw: torch.Tensor, tape: DeviceTape | None = None):
        """step() fed continuous action vectors ``raw[B, N, N+1]`` float32, converted on the device exactly as
        ``EdgeSimGym._convert_action`` does (MRRL.PY:283-305)."""
        B, n = self.n_envs, self.n_nodes
        if raw.dtype != torch.float32 or tuple(raw.shape) != (B, n, n + 1) or not raw.is_contiguous():
            raise ValueError(f"raw actions must be a contiguous float32 tensor of shape [{B}, {n}, {n + 1}]")
        if raw.device != self.device:
            raise ValueError("actions must live on the environment's device")
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_step_raw(self._h, raw.data_ptr(), C.byref(tape.struct) if tape is not None else None,
                                           C.byref(self._out), _stream_ptr(self.device)), "em_step_raw")
        return self._step_result()

    def step_greedy(self, resource_share: float = 0.5, tape: DeviceTape | None = None):
        """step() under the greedy baseline's policy (MRRL.PY:803-812): argmin(load/capacity), fixed share;
        the argmin is a wavefront-shuffle reduction inside the step kernel."""
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_step_greedy(self._h, float(resource_share),
                                              C.byref(tape.struct) if tape is not None else None, C.byref(self._out),
                                              _stream_ptr(self.device)), "em_step_greedy")
        return self._step_result()

    def step_many(self, k_steps: int, placement: torch.Tensor | None = None, resource: torch.Tensor | None = None,
                  raw: torch.Tensor | None = None, greedy_share: float | None = None, obs_history: bool = False):
        """``k_steps`` consecutive steps in ONE launch for open-loop actions (``em_step_many``): ``placement``/``resource``
        ``[K, B, N]``, or ``raw`` ``[K, B, N, N+1]``, or ``greedy_share`` for the greedy baseline's policy.  The state
        records stay on chip between the steps.  Returns ``(obs, reward[K, B] f64, done[K, B] u8, info)`` where ``obs`` is
        the last step's observation tensor, or ``[K, B, N, N+4]`` with ``obs_history=True`` (then ``info["active"]`` /
        ``info["rank"]`` carry the step dimension too); ``info["reward_raw"]``/``info["emitted"]`` are ``[K, B]``."""
        B, n, K, d = self.n_envs, self.n_nodes, int(k_steps), self.device
        modes = (placement is not None) + (raw is not None) + (greedy_share is not None)
        if modes != 1:
            raise ValueError("give exactly one of placement+resource, raw, greedy_share")
        pl_ptr = rs_ptr = None
        share = 0.5
        if placement is not None:
            if resource is None or placement.dtype != torch.int32 or resource.dtype != torch.float32 or \
                    tuple(placement.shape) != (K, B, n) or tuple(resource.shape) != (K, B, n) or \
                    not (placement.is_contiguous() and resource.is_contiguous()) or placement.device != d or resource.device != d:
                raise ValueError(f"placement (int32) and resource (float32) must be contiguous [{K}, {B}, {n}] tensors on {d}")
            mode, pl_ptr, rs_ptr = 0, placement.data_ptr(), resource.data_ptr()
        elif raw is not None:
            if raw.dtype != torch.float32 or tuple(raw.shape) != (K, B, n, n + 1) or not raw.is_contiguous() or raw.device != d:
                raise ValueError(f"raw actions must be a contiguous float32 [{K}, {B}, {n}, {n + 1}] tensor on {d}")
            mode, rs_ptr = 1, raw.data_ptr()
        else:
            mode, share = 2, float(greedy_share)
        f64, u8 = torch.float64, torch.uint8
        hist = {"reward": torch.zeros((K, B), dtype=f64, device=d), "reward_raw": torch.zeros((K, B), dtype=f64, device=d),
                "emitted": torch.zeros((K, B), dtype=u8, device=d), "done": torch.zeros((K, B), dtype=u8, device=d)}
        o = _lib.EmStepOut()
        o.reward_seen, o.reward = hist["reward"].data_ptr(), hist["reward_raw"].data_ptr()
        o.emitted, o.done = hist["emitted"].data_ptr(), hist["done"].data_ptr()
        history = 1
        if obs_history:
            obs = torch.empty((K, B, n, n + 4), dtype=torch.float32, device=d)
            active = torch.empty((K, B), dtype=torch.int64, device=d)
            rank = torch.empty((K, B, n), dtype=torch.int8, device=d)
            history |= 2
        else:
            obs, active, rank = self.obs, self.active, self.rank
        o.obs, o.active, o.rank = obs.data_ptr(), active.data_ptr(), rank.data_ptr()
        with torch.cuda.device(d):
            _lib.check(self._L.em_step_many(self._h, K, mode, pl_ptr, rs_ptr, share, history, C.byref(o), _stream_ptr(d)),
                       "em_step_many")
        if obs_history:   # keep the environment's own "current observation" buffers in step with the state
            self.obs.copy_(obs[-1]); self.active.copy_(active[-1]); self.rank.copy_(rank[-1])
        self.reward.copy_(hist["reward"][-1]); self.reward_raw.copy_(hist["reward_raw"][-1])
        self.emitted.copy_(hist["emitted"][-1]); self.done.copy_(hist["done"][-1])
        info = {"reward_raw": hist["reward_raw"], "emitted": hist["emitted"], "rank": rank, "active": active}
        return obs, hist["reward"], hist["done"], info

    def episode_metrics(self) -> torch.Tensor:
        out = torch.zeros((self.n_envs, 6), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_episode_metrics(self._h, out.data_ptr(), _stream_ptr(self.device)), "em_episode_metrics")
        return out

    def get_state(self) -> dict:
        B, n = self.n_envs, self.n_nodes
        st = {"load": np.zeros((B, n)), "queue": np.zeros((B, n)), "sim_time": np.zeros(B), "elapsed": np.zeros(B),
              "step": np.zeros(B, np.int32), "counts": np.zeros((B, 3), np.int64)}
        p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
        with torch.cuda.device(self.device):
            _lib.check(self._L.em_get_state(self._h, p(st["load"]), p(st["queue"]), p(st["sim_time"]), p(st["elapsed"]),
                                            p(st["step"]), p(st["counts"]), _stream_ptr(self.device)), "em_get_state")
        return st

    @property
    def nodes_load(self) -> np.ndarray:
        """[B, N] float64 host copy (the reference attribute read by the greedy baseline, MRRL.PY:808)."""
        return self.get_state()["load"]
