"""CPU restatement (numpy) of the PyMARL twin environment ``EdgeMARLEnv``.  TEST INFRASTRUCTURE ONLY.

Only ``tests/`` may import this module.  Restates ``pymarl/src/envs/edge_marl_env.py:9-318`` operation by operation in
float64.  Its dynamics differ from ``EdgeSimGym`` and must not be confused with it: every agent
holds a task every step (``:76-77,261-263``), the clock advances BEFORE the placements (``:83-85``), a placement adds the
task's workload to the target's load (``:127``), the reward uses the violation RATE (``:140,145``), the cyclic arrival
phase is shifted by -pi/2 (``:240``), divisions by capacity carry ``+ 1e-9`` (``:115,266,282``), and the generator is a
private ``numpy.random.Generator`` (``:45``).

Draw sources: ``GeneratorDraws`` (numpy ``default_rng``: stream-identical to the reference for a seed), ``FlatReplay``
(a log recorded from the reference's own generator), ``TapeDraws`` (the structured per-step tape the HIP kernel
consumes) and ``PhiloxDraws`` (the counter-based stream of the throughput mode with the deterministic log/sin of
``oracle/philox_ref.py``).  Pinned by ``tests/test_oracle_marl_env.py`` through ``tests/golden/marlenv_*.npz``, which
``oracle/make_golden_marl.py`` recorded from the imported reference.
"""
from __future__ import annotations

import math

import numpy as np

from . import philox_ref


def default_args() -> dict:
    """pymarl/src/config/envs/edge_marl.yaml:3-19."""
    return {
        "n_nodes": 10,
        "node_cpu_capacity": [1.0e9, 1.2e9, 1.4e9, 1.6e9, 1.8e9, 2.0e9, 2.2e9, 2.4e9, 2.6e9, 3.0e9],
        "task_arrival_rate": 10.0,
        "task_arrival_rate_range": [5.0, 15.0],
        "task_arrival_mode": "cyclic",
        "task_arrival_cycle_seconds": 20.0,
        "task_workload_range": [1.0, 10.0],
        "task_data_range": [0.5, 5.0],
        "task_deadline_range": [0.05, 0.5],
        "network_bw_range": [10.0, 100.0],
        "network_latency_range": [0.002, 0.01],
        "reward_weights": {"alpha": 0.5, "beta": 0.3, "gamma": 0.2},
        "min_time_step": 0.02,
        "max_time_step": 0.2,
        "initial_observation_window": 0.1,
        "action_mode": "discrete",
        "episode_length": 1000,
    }


class GeneratorDraws:
    def __init__(self, seed=None):
        self.rng = np.random.default_rng(seed)

    def begin_step(self, s):
        pass

    def rate_u(self, lo, hi):
        return self.rng.uniform(lo, hi)

    def dt_raw(self, scale):
        return self.rng.exponential(scale)

    def net(self, agent, bw_rng, lat_rng):
        return self.rng.uniform(*bw_rng), self.rng.uniform(*lat_rng)

    def task(self, agent, wl_rng, data_rng, ddl_rng):
        return self.rng.uniform(*wl_rng), self.rng.uniform(*data_rng), self.rng.uniform(*ddl_rng)


class FlatReplay:
    """Replays ``(name, args, result)`` entries recorded from the reference's generator and checks that the
    restatement asks for the same distribution with identical arguments in the same order."""

    def __init__(self, log):
        self.log, self.pos = log, 0

    def begin_step(self, s):
        pass

    def _take(self, name, *args):
        rname, rargs, rval = self.log[self.pos]
        got = tuple(float(a) for a in args)
        if rname != name or tuple(rargs) != got:
            raise AssertionError(f"draw #{self.pos}: reference {rname}{tuple(rargs)} vs restatement {name}{got}")
        self.pos += 1
        return rval

    def rate_u(self, lo, hi):
        return self._take("uniform", lo, hi)

    def dt_raw(self, scale):
        return self._take("exponential", scale)

    def net(self, agent, bw_rng, lat_rng):
        return self._take("uniform", *bw_rng), self._take("uniform", *lat_rng)

    def task(self, agent, wl_rng, data_rng, ddl_rng):
        return self._take("uniform", *wl_rng), self._take("uniform", *data_rng), self._take("uniform", *ddl_rng)


class TapeDraws:
    def __init__(self, tape):
        self.tape, self.s = tape, 0

    def begin_step(self, s):
        self.s = s

    def rate_u(self, lo, hi):
        return float(self.tape["rate_u"][self.s])

    def dt_raw(self, scale):
        return float(self.tape["dt_raw"][self.s])

    def net(self, agent, bw_rng, lat_rng):
        return float(self.tape["bw_u"][self.s, agent]), float(self.tape["lat"][self.s, agent])

    def task(self, agent, wl_rng, data_rng, ddl_rng):
        t = self.tape
        return float(t["task_wl"][self.s, agent]), float(t["task_data"][self.s, agent]), float(t["task_ddl_off"][self.s, agent])


class PhiloxDraws:
    """Counter-based stream of the throughput mode: (seed; env, episode, step, purpose, agent)."""

    det_math = True

    def __init__(self, seed, env_index, episode=1):
        self.seed, self.env, self.episode, self.s = int(seed), int(env_index), int(episode), 0

    def begin_step(self, s):
        self.s = s

    def _blk(self, purpose, idx):
        return philox_ref.block(self.seed, self.env, self.episode, self.s, purpose, idx)

    def rate_u(self, lo, hi):
        return philox_ref.uniform(philox_ref.d1(self._blk(philox_ref.P_DT, 0)), lo, hi)

    def dt_raw(self, scale):
        return scale * (-philox_ref.det_log(1.0 - philox_ref.d0(self._blk(philox_ref.P_DT, 0))))

    def net(self, agent, bw_rng, lat_rng):
        b = self._blk(philox_ref.P_NET, agent)
        return philox_ref.uniform(philox_ref.d0(b), *bw_rng), philox_ref.uniform(philox_ref.d1(b), *lat_rng)

    def task(self, agent, wl_rng, data_rng, ddl_rng):
        a, b = self._blk(philox_ref.P_TASK_A, agent), self._blk(philox_ref.P_TASK_B, agent)
        return (philox_ref.uniform(philox_ref.d0(a), *wl_rng), philox_ref.uniform(philox_ref.d1(a), *data_rng),
                philox_ref.uniform(philox_ref.d0(b), *ddl_rng))


def empty_tape(n_rows, n):
    return {"rate_u": np.full(n_rows, np.nan), "dt_raw": np.full(n_rows, np.nan), "bw_u": np.full((n_rows, n), np.nan),
            "lat": np.full((n_rows, n), np.nan), "task_wl": np.zeros((n_rows, n)), "task_data": np.zeros((n_rows, n)),
            "task_ddl_off": np.zeros((n_rows, n))}


def convert_hybrid_action(vec, n):
    """``_convert_hybrid_action`` (edge_marl_env.py:304-318): (placement, float64 resource share)."""
    vec = np.asarray(vec, dtype=np.float32).reshape(-1)
    if vec.size < n + 1:
        padded = np.zeros(n + 1, dtype=np.float32)
        padded[:vec.size] = vec
        vec = padded
    logits = vec[:n]
    place = 0 if np.allclose(logits, 0.0) else int(np.argmax(logits)) % n
    norm = (float(vec[-1]) + 1.0) / 2.0
    return place, 0.1 + 0.9 * np.clip(norm, 0.0, 1.0)


class EdgeMARLOracle:
    def __init__(self, args=None, draws=None, tape_out=None):
        a = default_args() if args is None else args
        self.n = int(a.get("n_nodes", 10))
        self.limit = int(a.get("episode_length", 1000))
        self.cap = np.asarray(a.get("node_cpu_capacity", np.ones(self.n) * 1e9), dtype=np.float64)
        self.wl_rng = tuple(a.get("task_workload_range", (1.0, 10.0)))
        self.data_rng = tuple(a.get("task_data_range", (0.5, 5.0)))
        self.ddl_rng = tuple(a.get("task_deadline_range", (0.05, 0.5)))
        self.bw_rng = tuple(a.get("network_bw_range", (10.0, 100.0)))
        self.lat_rng = tuple(a.get("network_latency_range", (0.002, 0.01)))
        self.w = dict(a.get("reward_weights", {"alpha": 0.5, "beta": 0.3, "gamma": 0.2}))
        self.rate = a.get("task_arrival_rate", 10.0)
        self.rate_range = tuple(a.get("task_arrival_rate_range", (5.0, 15.0)))
        self.rate_mode = a.get("task_arrival_mode", "cyclic")
        self.cycle = a.get("task_arrival_cycle_seconds", 20.0)
        self.dt_min, self.dt_max = a.get("min_time_step", 0.02), a.get("max_time_step", 0.2)
        self.mode = a.get("action_mode", "discrete")
        self.draws = draws if draws is not None else GeneratorDraws(a.get("seed"))
        self.tape_out = tape_out
        self.reset()

    def _rec(self, key, val, agent=None):
        if self.tape_out is None:
            return
        if agent is None:
            self.tape_out[key][self.step_no] = val
        else:
            self.tape_out[key][self.step_no, agent] = val

    # edge_marl_env.py:245-254
    def _sample_tasks(self):
        self.wl, self.data, self.deadline = np.zeros(self.n), np.zeros(self.n), np.zeros(self.n)
        for i in range(self.n):
            wl, data, off = self.draws.task(i, self.wl_rng, self.data_rng, self.ddl_rng)
            self._rec("task_wl", wl, i); self._rec("task_data", data, i); self._rec("task_ddl_off", off, i)
            self.wl[i], self.data[i], self.deadline[i] = wl, data, self.sim_time + off

    # :67-77
    def reset(self):
        self.step_no, self.sim_time = 0, 0.0
        self.load, self.queue = np.zeros(self.n), np.zeros(self.n)
        self.metrics = {k: [] for k in ("latency", "energy", "violations", "throughput", "load_balance")}
        self.draws.begin_step(0)
        self._sample_tasks()

    # :236-243
    def _rate(self):
        if self.rate_mode == "cyclic":
            phase = (self.sim_time % self.cycle) / self.cycle
            lo, hi = self.rate_range
            if getattr(self.draws, "det_math", False):     # throughput mode: build-defined sine (sin(x - pi/2) at phase - 1/4)
                return lo + (hi - lo) * 0.5 * (1 + philox_ref.det_sin2pi(phase - 0.25))
            return lo + (hi - lo) * 0.5 * (1 + math.sin(2 * math.pi * phase - math.pi / 2))
        if self.rate_mode == "random":
            u = self.draws.rate_u(*self.rate_range)
            self._rec("rate_u", u)
            return u
        return self.rate

    # :79-162
    def step(self, actions):
        self.step_no += 1
        self.draws.begin_step(self.step_no)
        rate = self._rate()
        raw = self.draws.dt_raw(1.0 / max(rate, 1e-6))
        self._rec("dt_raw", raw)
        window = float(np.clip(raw, self.dt_min, self.dt_max))
        self.sim_time += window
        lats, ens, viols = [], [], []
        targets = []
        for i in range(self.n):
            if self.mode == "discrete":
                target, share = int(np.clip(actions[i], 0, self.n - 1)), 0.5
            else:
                target, share = convert_hybrid_action(actions[i], self.n)
            load_t = self.load[target]
            if target == i:
                transfer = 0.0
            else:
                bw_u, lat = self.draws.net(i, self.bw_rng, self.lat_rng)
                self._rec("bw_u", bw_u, i); self._rec("lat", lat, i)
                transfer = (self.data[i] / (bw_u * 1e6 / 8.0)) + lat
            queue_lat = load_t / (self.cap[target] + 1e-9)
            eff = self.cap[target] * share
            exec_lat = self.wl[i] / (eff / 1e9)
            total = transfer + queue_lat + exec_lat
            lats.append(total)
            ens.append(self.wl[i] * (eff / 1e9))
            viols.append(1 if (self.sim_time + total) > self.deadline[i] else 0)
            self.load[target] += self.wl[i]
            self.queue[target] += 1
            targets.append(target)
        self.load = np.maximum(0.0, self.load - self.cap * window)      # :256-259
        self.queue = np.maximum(0.0, self.queue - 1)
        self._sample_tasks()
        avg_lat, avg_en, viol_rate = float(np.mean(lats)), float(np.mean(ens)), float(np.mean(viols))
        throughput = (self.n / window) if window > 0 else 0.0
        vals = self.load + 1e-6
        den = vals.size * np.sum(np.square(vals))
        jain = 0.0 if den <= 0 else float(np.square(np.sum(vals)) / den)
        reward = -(self.w.get("alpha", 0.5) * avg_lat + self.w.get("beta", 0.3) * avg_en + self.w.get("gamma", 0.2) * viol_rate)
        terminated = self.step_no >= self.limit
        for k, v in zip(("latency", "energy", "violations", "throughput", "load_balance"), (avg_lat, avg_en, viol_rate, throughput, jain)):
            self.metrics[k].append(v)
        info = {"avg_latency_ms": avg_lat * 1000.0, "avg_energy_J": avg_en, "deadline_violation_rate": viol_rate,
                "throughput_tps": throughput, "load_balance_jain": jain, "episode_limit": terminated}
        self.last = {"latencies": lats, "targets": targets, "window": window}
        return reward, terminated, info

    # :265-288
    def obs(self):
        ratio = self.load / (self.cap + 1e-9)
        out = np.zeros((self.n, self.n + 4), np.float32)
        for i in range(self.n):
            head = [ratio[i], self.queue[i], self.data[i], self.wl[i], max(0.0, self.deadline[i] - self.sim_time)]
            out[i] = np.array(head + [ratio[j] for j in range(self.n) if j != i], dtype=np.float32)
        return out

    # :193-207
    def episode_summary(self):
        m = {k: np.array(v, dtype=np.float32) for k, v in self.metrics.items()}
        lat_ms = m["latency"] * 1000.0
        return {"avg_latency_ms": float(lat_ms.mean()) if lat_ms.size else 0.0,
                "p99_latency_ms": float(np.percentile(lat_ms, 99)) if lat_ms.size else 0.0,
                "avg_energy_J": float(m["energy"].mean()) if lat_ms.size else 0.0,
                "throughput_tps": float(m["throughput"].mean()) if lat_ms.size else 0.0,
                "deadline_violation_rate": float(m["violations"].mean()) if lat_ms.size else 0.0,
                "load_balance_jain": float(m["load_balance"].mean()) if lat_ms.size else 0.0}
