"""EdgeMARLEnv: the PyMARL twin environment (pymarl/src/envs/edge_marl_env.py), batched over ``n_envs`` on an MI355X.

Same constructor keys (``env_args`` of config/envs/edge_marl.yaml), same ``MultiAgentEnv`` method names
(pymarl/src/envs/multiagentenv.py) and the same arithmetic; every returned quantity gains a leading ``n_envs``
dimension and stays on the device.  ``csrc/edge_marl_env.hip`` does the work; there is no CPU path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .config import MARL_ENV_ARGS, make_emm_config
from .env import _stream_ptr

INFO_KEYS = ("avg_latency_ms", "avg_energy_J", "deadline_violation_rate", "throughput_tps", "load_balance_jain")
_TAPE_FIELDS = ("rate_u", "dt_raw", "bw_u", "net_lat", "task_workload", "task_data", "task_deadline_offset")


class MarlDeviceTape:
    """One step of pre-drawn random numbers on the device (``emm_tape``); float64 arrays with leading dimension n_envs."""

    def __init__(self, arrays: dict, device):
        self.struct = _lib.EmmTape()
        self._keep = []
        for name in _TAPE_FIELDS:
            if arrays.get(name) is None:
                continue
            t = torch.as_tensor(np.ascontiguousarray(np.nan_to_num(np.asarray(arrays[name], dtype=np.float64), nan=0.0)), device=device)
            self._keep.append(t)
            setattr(self.struct, name, t.data_ptr())


def episode_summary_from_log(log: np.ndarray) -> dict:
    """``get_episode_summary`` (edge_marl_env.py:193-207) for every environment.  ``log`` is ``[steps, n_envs, 6]`` float64:
    the five info scalars plus the step's mean latency in seconds.  float32 arithmetic and numpy's reduction order, as
    the reference (each environment's series is reduced along a contiguous axis)."""
    steps, n_envs = log.shape[0], log.shape[1]
    keys = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "deadline_violation_rate", "load_balance_jain")
    if steps == 0:
        return {k: np.zeros(n_envs) for k in keys}
    series = np.ascontiguousarray(np.transpose(log, (2, 1, 0)).astype(np.float32))       # [6][n_envs][steps]
    latency_ms = series[5] * 1000.0
    mean = lambda a: np.array([float(row.mean()) for row in a])                           # noqa: E731
    return {
        "avg_latency_ms": mean(latency_ms),
        "p99_latency_ms": np.array([float(np.percentile(row, 99)) for row in latency_ms]),
        "avg_energy_J": mean(series[1]),
        "throughput_tps": mean(series[3]),
        "deadline_violation_rate": mean(series[2]),
        "load_balance_jain": mean(series[4]),
    }


class EdgeMARLEnv:
    """reset() ; step(actions) -> (reward[B] f64, terminated[B] bool, info dict of [B] f64 tensors).

    ``actions``: ``[B, n]`` integer tensor in discrete mode (clipped to ``[0, n-1]``, edge_marl_env.py:101), ``[B, n, n+1]``
    float32 in hybrid mode (:304-318).  Without a tape the environments draw from the in-kernel Philox stream
    ``(seed; env_offset + b, episode, step)``; the reference's own generator stream is reproduced through tapes."""

    def __init__(self, n_envs: int = 1, device="cuda:0", env_offset: int = 0, **env_args):
        self.args = dict(MARL_ENV_ARGS)
        self.args.update(env_args)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("EdgeMARLEnv runs on an MI355X (device='cuda:N'); there is no CPU path")
        self._cfg = make_emm_config(self.args)
        self.n_envs = int(n_envs)
        self.n_agents = int(self._cfg.n_nodes)
        self.episode_limit = int(self._cfg.episode_length)
        self.action_mode = self.args.get("action_mode", "discrete")
        self.hybrid_action_dim = self.n_agents + 1
        self.obs_dim = 5 + (self.n_agents - 1)
        self.state_dim = self.obs_dim * self.n_agents
        self.random_seed = self.args.get("seed", None)
        self._L = _lib.lib()
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._h = C.c_void_p()
        seed = 0 if self.random_seed is None else int(self.random_seed)
        with torch.cuda.device(dev_index):
            _lib.check(self._L.emm_create(C.byref(self._cfg), self.n_envs, dev_index, seed & (2**64 - 1), int(env_offset),
                                          C.byref(self._h)), "emm_create")
        B, n, d = self.n_envs, self.n_agents, self.device
        self.obs = torch.zeros((B, n, n + 4), dtype=torch.float32, device=d)
        self.reward = torch.zeros(B, dtype=torch.float64, device=d)
        self.terminated = torch.zeros(B, dtype=torch.bool, device=d)        # the kernel stores 0/1 bytes
        self.latency = None
        self._log = torch.zeros((self.episode_limit, B, 6), dtype=torch.float64, device=d)
        self._avail = torch.ones((B, n, n), dtype=torch.int32, device=d)
        self._out = _lib.EmmStepOut()
        self._out.obs, self._out.reward, self._out.terminated = self.obs.data_ptr(), self.reward.data_ptr(), self.terminated.data_ptr()
        self.current_step = 0
        self._pending_seed = None
        self.reset()                                                                       # edge_marl_env.py:58

    def enable_debug_outputs(self):
        self.latency = torch.zeros((self.n_envs, self.n_agents), dtype=torch.float64, device=self.device)
        self._out.latency = self.latency.data_ptr()

    def set_lanes_per_env(self, lanes_per_env: int):
        """Kernel variant: 1, 16 or 0 (automatic); tuning/tests only, identical results."""
        _lib.check(self._L.emm_set_lanes_per_env(self._h, int(lanes_per_env)), "emm_set_lanes_per_env")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.emm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def seed(self, seed: int):
        """edge_marl_env.py:60-62: takes effect at the next reset."""
        self.random_seed = seed
        self._pending_seed = int(seed)

    # ---------------------------------------------------------------- MultiAgentEnv interface
    def reset(self, tape: MarlDeviceTape | None = None):
        has_seed, seed = (0, 0) if self._pending_seed is None else (1, self._pending_seed & (2**64 - 1))
        self._pending_seed = None
        self._out.info = None
        with torch.cuda.device(self.device):
            _lib.check(self._L.emm_reset(self._h, C.byref(tape.struct) if tape is not None else None, has_seed, seed,
                                         C.byref(self._out), _stream_ptr(self.device)), "emm_reset")
        self.current_step = 0

    def step(self, actions: torch.Tensor, tape: MarlDeviceTape | None = None):
        B, n = self.n_envs, self.n_agents
        if not isinstance(actions, torch.Tensor) or actions.device != self.device:
            raise ValueError("actions must be a tensor on the environment's device")
        if self.action_mode == "discrete":
            if tuple(actions.shape) != (B, n):
                raise ValueError("Actions must be provided for each agent.")              # edge_marl_env.py:80-81
            if actions.dtype != torch.int32:
                actions = actions.to(torch.int32)
            a_i, a_f = actions.contiguous(), None
        else:
            if tuple(actions.shape) != (B, n, n + 1):
                raise ValueError("Actions must be provided for each agent.")
            a_i, a_f = None, actions.to(torch.float32).contiguous()
        if self.current_step >= self._log.shape[0]:                                        # stepping past the limit, as the reference allows
            self._log = torch.cat([self._log, torch.zeros_like(self._log)], dim=0)
        row = self._log[self.current_step]
        self._out.info = row.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(self._L.emm_step(self._h, a_i.data_ptr() if a_i is not None else None,
                                        a_f.data_ptr() if a_f is not None else None,
                                        C.byref(tape.struct) if tape is not None else None, C.byref(self._out),
                                        _stream_ptr(self.device)), "emm_step")
        self.current_step += 1
        info = {k: row[:, j] for j, k in enumerate(INFO_KEYS)}
        info["episode_limit"] = self.terminated                                            # edge_marl_env.py:160
        return self.reward, self.terminated, info

    def get_obs(self):
        return self.obs

    def get_obs_agent(self, agent_id):
        return self.obs[:, agent_id]

    def get_obs_size(self):
        return self.obs_dim

    def get_state(self):
        return self.obs.view(self.n_envs, self.state_dim)                                  # concat of the rows, :173-174

    def get_state_size(self):
        return self.state_dim

    def get_avail_actions(self):
        return self._avail

    def get_avail_agent_actions(self, agent_id):
        return self._avail[:, agent_id]

    def get_total_actions(self):
        return self.n_agents

    def get_action_dim(self):
        return 1 if self.action_mode == "discrete" else self.hybrid_action_dim

    def get_episode_summary(self):
        """Dict of ``[n_envs]`` float64 arrays (edge_marl_env.py:193-207 per environment)."""
        return episode_summary_from_log(self._log[:self.current_step].cpu().numpy())

    def get_env_info(self):
        """multiagentenv.py:54-60."""
        return {"state_shape": self.get_state_size(), "obs_shape": self.get_obs_size(), "n_actions": self.get_total_actions(),
                "n_agents": self.n_agents, "episode_limit": self.episode_limit}

    def render(self):
        return

    def save_replay(self):
        return

    # ---------------------------------------------------------------- tests
    def raw_state(self) -> dict:
        B, n, d = self.n_envs, self.n_agents, self.device
        st = {"load": torch.zeros((B, n), dtype=torch.float64, device=d), "queue": torch.zeros((B, n), dtype=torch.float64, device=d),
              "sim_time": torch.zeros(B, dtype=torch.float64, device=d), "step": torch.zeros(B, dtype=torch.int32, device=d)}
        with torch.cuda.device(self.device):
            _lib.check(self._L.emm_get_state(self._h, st["load"].data_ptr(), st["queue"].data_ptr(), st["sim_time"].data_ptr(),
                                             st["step"].data_ptr(), _stream_ptr(self.device)), "emm_get_state")
        return {k: v.cpu().numpy() for k, v in st.items()}


class EdgeMARLEnvHost:
    """One environment with the reference's host-side types: what ``REGISTRY["edge_marl"]`` (pymarl/src/envs/__init__.py:20)
    binds for PyMARL's episode runner, which steps one environment per process and expects a float reward, a bool flag,
    a dict of floats, a list of numpy observation rows and a numpy state.  Every call is one kernel launch plus one small
    device-to-host copy; batched collectors should use ``EdgeMARLEnv`` and keep everything on the device."""

    def __init__(self, device="cuda:0", **env_args):
        self._env = EdgeMARLEnv(n_envs=1, device=device, **env_args)
        for name in ("n_agents", "episode_limit", "action_mode", "hybrid_action_dim", "obs_dim", "state_dim"):
            setattr(self, name, getattr(self._env, name))

    def seed(self, seed: int):
        self._env.seed(seed)

    def reset(self):
        self._env.reset()

    def step(self, actions):
        n = self.n_agents
        if len(actions) != n:
            raise ValueError("Actions must be provided for each agent.")                   # edge_marl_env.py:80-81
        rows = [a.detach().cpu().numpy() if hasattr(a, "cpu") else np.asarray(a) for a in actions]
        if self.action_mode == "discrete":
            vals = np.array([np.clip(r, 0, n - 1).reshape(-1)[0] for r in rows])          # int(np.clip(...)), :101
            a = torch.as_tensor(vals.astype(np.int32)[None], device=self._env.device)
        else:
            mat = np.zeros((n, n + 1), dtype=np.float32)
            for i, r in enumerate(rows):                                                  # zero padding of short vectors, :305-309
                v = np.asarray(r, dtype=np.float32).reshape(-1)
                if v.size > n + 1:
                    v = np.concatenate([v[:n], v[-1:]])                                   # logits = the first n, resource = the last
                mat[i, :v.size] = v
            a = torch.as_tensor(mat[None], device=self._env.device)
        reward, term, info = self._env.step(a)
        host = torch.stack([info[k][0] for k in INFO_KEYS] + [reward[0]]).cpu().numpy()
        terminated = bool(term[0].item())
        out = {k: float(host[j]) for j, k in enumerate(INFO_KEYS)}
        out["episode_limit"] = terminated
        return float(host[-1]), terminated, out

    def get_obs(self):
        o = self._env.get_obs()[0].cpu().numpy()
        return [o[i] for i in range(self.n_agents)]

    def get_obs_agent(self, agent_id):
        return self._env.get_obs()[0, agent_id].cpu().numpy()

    def get_obs_size(self):
        return self.obs_dim

    def get_state(self):
        return self._env.get_state()[0].cpu().numpy()

    def get_state_size(self):
        return self.state_dim

    def get_avail_actions(self):
        return np.ones((self.n_agents, self.get_total_actions()), dtype=np.int32)

    def get_avail_agent_actions(self, agent_id):
        return np.ones(self.get_total_actions(), dtype=np.int32)

    def get_total_actions(self):
        return self.n_agents

    def get_action_dim(self):
        return self._env.get_action_dim()

    def get_episode_summary(self):
        return {k: float(v[0]) for k, v in self._env.get_episode_summary().items()}

    def get_env_info(self):
        return self._env.get_env_info()

    def close(self):
        self._env.close()

    def render(self):
        return

    def save_replay(self):
        return
