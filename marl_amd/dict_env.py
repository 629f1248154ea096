"""Dict-API view (n_envs = 1) of the batched environment: the reference's ``EdgeSimGym`` call surface
(``reset(*, seed=None, options=None)`` / ``step(action_dict)`` and the attributes its callers read,
MRRL.PY:114-521) for scripts that drive a single environment agent by agent."""
from __future__ import annotations

import numpy as np
import torch

from .config import ENV_CONFIG
from .env import METRIC_KEYS, BatchedEdgeSimGym


class EdgeSimGym:
    # the reference keeps the exploration step counter on the class, shared by every instance (MRRL.PY:122)
    _global_step_counter = 0

    def __init__(self, config=ENV_CONFIG, device="cuda:0", seed=0):
        self._env = BatchedEdgeSimGym(config, n_envs=1, device=device, seed=seed)
        self.n_nodes = self._env.n_nodes
        self.episode_len = self._env.episode_len
        self.cpu_caps = self._env.cpu_caps
        self._agent_ids = set(f"node_{i}" for i in range(self.n_nodes))
        n = self.n_nodes
        self._placement = torch.zeros((1, n), dtype=torch.int32, device=self._env.device)
        self._resource = torch.zeros((1, n), dtype=torch.float32, device=self._env.device)
        self.current_step = 0

    # -- attributes read/written by callers --------------------------------------------------------------
    @property
    def disable_env_exploration(self):
        return self._env.disable_env_exploration

    @disable_env_exploration.setter
    def disable_env_exploration(self, v):
        self._env.disable_env_exploration = bool(v)

    @property
    def nodes_load(self):
        return self._env.get_state()["load"][0]

    @property
    def nodes_queue_len(self):
        return self._env.get_state()["queue"][0]

    @property
    def sim_time(self):
        return float(self._env.get_state()["sim_time"][0])

    # -- helpers ---------------------------------------------------------------------------------------------
    def _obs_dict(self):
        obs = self._env.obs[0].cpu().numpy()
        rank = self._env.rank[0].cpu().numpy()
        order = sorted((int(r), a) for a, r in enumerate(rank) if r >= 0)
        return {f"node_{a}": obs[a].copy() for _, a in order}

    def reset(self, *, seed=None, options=None):
        self._env.reset(seed=seed)
        self.current_step = 0
        obs = self._obs_dict()
        return obs, {k: {} for k in obs}

    def step(self, action_dict):
        n = self.n_nodes
        env = self._env
        if len(action_dict) == n and all(not isinstance(a, dict) and np.size(a) == n + 1 for a in action_dict.values()):
            # every agent supplied a full continuous vector: converted on the device (em_step_raw)
            raw = np.zeros((1, n, n + 1), np.float32)
            for name, act in action_dict.items():
                raw[0, int(name.split("_")[1])] = np.asarray(act, np.float32).reshape(-1)
            env.global_step_counter = EdgeSimGym._global_step_counter
            env.step_raw(torch.from_numpy(raw).to(env.device))
            return self._after_step()
        pl = np.full((1, n), -1, np.int32)       # agents without an entry are not dispatched (MRRL.PY:397-399)
        rs = np.full((1, n), 0.5, np.float32)
        for name, act in action_dict.items():
            a = int(name.split("_")[1])
            if isinstance(act, dict):
                pl[0, a] = int(act["placement"])
                rs[0, a] = np.float32(np.asarray(act["resource"]).reshape(-1)[0])
            else:                                   # continuous vector form, _convert_action (MRRL.PY:283-305)
                vec = np.zeros(n + 1, np.float32)
                v = np.asarray(act, np.float32).reshape(-1)[: n + 1]
                vec[: v.size] = v
                pl[0, a] = 0 if np.allclose(vec[:-1], 0) else int(np.argmax(vec[:-1])) % n
                rs[0, a] = np.float32(0.1 + 0.9 * np.clip((float(vec[-1]) + 1.0) / 2.0, 0.0, 1.0))
        self._placement.copy_(torch.from_numpy(pl))
        self._resource.copy_(torch.from_numpy(rs))
        env.global_step_counter = EdgeSimGym._global_step_counter
        env.step(self._placement, self._resource)
        return self._after_step()

    def _after_step(self):
        env = self._env
        EdgeSimGym._global_step_counter = env.global_step_counter
        self.current_step += 1
        obs = self._obs_dict()
        reward = float(env.reward_raw[0].item())
        done = bool(env.done[0].item())
        rewards = {k: reward for k in obs}
        infos = {k: {} for k in obs}
        if done:
            m = env.episode_metrics()[0].cpu().numpy()
            metrics = {k: float(m[i]) for i, k in enumerate(METRIC_KEYS)}
            for k in (list(obs.keys()) or ["__common__"]):
                info = infos.setdefault(k, {})
                info["episode_metrics"] = dict(metrics)
                info["custom_metrics"] = dict(metrics)
                info.update(metrics)
        return obs, rewards, {"__all__": done}, {"__all__": False}, infos
