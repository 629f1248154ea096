"""CPU restatement (numpy) of the reference edge-computing simulator.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product (``marl_amd``) never does.

Restates ``EdgeSimGym`` (reference ``MRRL.PY:114-521``) operation by operation in float64, in the
reference's evaluation order, but organised around a *draw source* so the same arithmetic can be
driven by

* ``LegacyDraws``  - numpy's legacy ``RandomState`` (bit-identical to the reference for a seed),
* ``FlatReplay``   - a flat log of draws captured from the reference (``oracle/ref_loader.py``),
* ``TapeDraws``    - the structured per-step tape that the HIP kernels consume in parity mode
                     (SURVEY.md appendix C),
* ``PhiloxDraws``  - the counter-based stream of the throughput mode (``oracle/philox_ref.py``).

Every run also *emits* the structured tape of what it drew, which is how golden tapes are made.

Pinned against the reference by ``tests/test_oracle_env.py`` through the fixtures under
``tests/golden/`` that ``oracle/make_golden.py`` wrote from the imported reference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

from . import philox_ref

A_SLOTS = 16  # arrival slots per tape step (reference max observed 9, Poisson mean <= 3)


# ----------------------------------------------------------------------------------------------
# configuration (values of MRRL.PY:70-110, restated)
# ----------------------------------------------------------------------------------------------
def default_config() -> dict:
    return {
        "n_nodes": 10,
        "task_arrival_rate": 10,
        "task_arrival_rate_range": (5, 15),
        "task_arrival_mode": "cyclic",
        "task_arrival_cycle_seconds": 20.0,
        "episode_length": 1000,
        "node_cpu_capacity": np.array([1.0, 1.2, 1.4, 1.6, 1.8, 2.0, 2.2, 2.4, 2.6, 3.0]) * 1e9,
        "task_workload_range": (1.0, 10.0),
        "task_data_range": (0.5, 5.0),
        "task_deadline_range": (0.05, 0.5),
        "network_bw_range": (10, 100),
        "network_latency_range": (0.002, 0.01),
        "reward_weights": {"alpha": 0.5, "beta": 0.3, "gamma": 0.2},
        "min_time_step": 0.02,
        "max_time_step": 0.2,
        "initial_observation_window": 0.1,
        "exploration_schedule": {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 1_000_000},
        "disable_env_exploration": False,
    }


# ----------------------------------------------------------------------------------------------
# draw sources.  Each method names the PURPOSE of the draw so structured sources can serve it.
# ----------------------------------------------------------------------------------------------
class LegacyDraws:
    """numpy legacy RandomState; the reference uses the module-global one (MRRL.PY:189-191)."""

    def __init__(self, seed=None, state=None):
        self.rs = state if state is not None else np.random.RandomState(seed)

    def reseed(self, seed):
        self.rs.seed(seed)

    def begin_step(self, step_index):
        pass

    def rate_u(self, lo, hi):
        return self.rs.uniform(lo, hi)

    def dt_raw(self, scale):
        return self.rs.exponential(scale)

    def n_arrivals(self, lam):
        return int(self.rs.poisson(lam))

    def task(self, j, wl_rng, data_rng, ddl_rng, n):
        wl = self.rs.uniform(*wl_rng)
        data = self.rs.uniform(*data_rng)
        off = self.rs.uniform(*ddl_rng)
        dec = int(self.rs.randint(0, n))
        return wl, data, off, dec

    def explore_u(self, k, agent):
        return self.rs.rand()

    def explore_action(self, k, agent, n):
        tgt = int(self.rs.randint(0, n))
        res = self.rs.uniform(0.1, 1.0)
        return tgt, res

    def net(self, k, agent, bw_rng, lat_rng):
        bw_u = self.rs.uniform(*bw_rng)
        lat = self.rs.uniform(*lat_rng)
        return bw_u, lat


class FlatReplay:
    """Replays a flat ``(name, args, result)`` log captured from the reference and checks that the
    restatement asks for the same distribution with bit-identical arguments in the same order."""

    def __init__(self, log):
        self.log = log
        self.pos = 0

    def reseed(self, seed):
        pass

    def begin_step(self, step_index):
        pass

    def _take(self, name, *args):
        if self.pos >= len(self.log):
            raise AssertionError(f"restatement draws more than the reference did (at {name}{args})")
        rname, rargs, rval = self.log[self.pos]
        self.pos += 1
        got = tuple(float(a) for a in args)
        if rname != name or tuple(rargs) != got:
            raise AssertionError(f"draw #{self.pos - 1}: reference {rname}{tuple(rargs)} vs restatement {name}{got}")
        return rval

    def rate_u(self, lo, hi):
        return self._take("uniform", lo, hi)

    def dt_raw(self, scale):
        return self._take("exponential", scale)

    def n_arrivals(self, lam):
        return int(self._take("poisson", lam))

    def task(self, j, wl_rng, data_rng, ddl_rng, n):
        wl = self._take("uniform", *wl_rng)
        data = self._take("uniform", *data_rng)
        off = self._take("uniform", *ddl_rng)
        dec = int(self._take("randint", 0, n))
        return wl, data, off, dec

    def explore_u(self, k, agent):
        return self._take("rand")

    def explore_action(self, k, agent, n):
        tgt = int(self._take("randint", 0, n))
        res = self._take("uniform", 0.1, 1.0)
        return tgt, res

    def net(self, k, agent, bw_rng, lat_rng):
        return self._take("uniform", *bw_rng), self._take("uniform", *lat_rng)


class TapeDraws:
    """Serves draws from a structured tape (dict of arrays with leading step axis)."""

    def __init__(self, tape):
        self.tape = tape
        self.s = 0

    def reseed(self, seed):
        pass

    def begin_step(self, step_index):
        self.s = step_index

    @staticmethod
    def _need(v, what):
        if isinstance(v, float) and math.isnan(v):
            raise AssertionError(f"tape has no draw for {what}: the run diverged from the recording")
        return v

    def rate_u(self, lo, hi):
        return self._need(float(self.tape["rate_u"][self.s]), "rate_u")

    def dt_raw(self, scale):
        return self._need(float(self.tape["dt_raw"][self.s]), "dt_raw")

    def n_arrivals(self, lam):
        return int(self.tape["n_arr"][self.s])

    def task(self, j, wl_rng, data_rng, ddl_rng, n):
        t = self.tape
        return (float(t["task_wl"][self.s, j]), float(t["task_data"][self.s, j]),
                float(t["task_ddl_off"][self.s, j]), int(t["task_decider"][self.s, j]))

    def explore_u(self, k, agent):
        return self._need(float(self.tape["explore_u"][self.s, k]), "explore_u")

    def explore_action(self, k, agent, n):
        return int(self.tape["explore_tgt"][self.s, k]), self._need(float(self.tape["explore_res"][self.s, k]), "explore_res")

    def net(self, k, agent, bw_rng, lat_rng):
        return (self._need(float(self.tape["bw_u"][self.s, k]), "bw_u"),
                self._need(float(self.tape["lat"][self.s, k]), "lat"))


class PhiloxDraws:
    """Counter-based stream of the throughput mode; must mirror csrc/edge_rng.h exactly."""

    det_math = True

    def __init__(self, seed, env_index, episode=0):
        self.seed = int(seed)
        self.env = int(env_index)
        self.episode = int(episode)
        self.s = 0
        self._pois_lam = None

    def reseed(self, seed):
        self.seed = int(seed)
        self.episode = 0

    def begin_step(self, step_index):
        self.s = step_index

    def _blk(self, purpose, idx):
        return philox_ref.block(self.seed, self.env, self.episode, self.s, purpose, idx)

    def rate_u(self, lo, hi):
        return philox_ref.uniform(philox_ref.d1(self._blk(philox_ref.P_DT, 0)), lo, hi)

    def dt_raw(self, scale):
        u = philox_ref.d0(self._blk(philox_ref.P_DT, 0))
        return scale * (-philox_ref.det_log(1.0 - u))

    def n_arrivals(self, lam):
        return philox_ref.poisson_mult(lambda i: self._blk(philox_ref.P_POIS, i), lam)

    def task(self, j, wl_rng, data_rng, ddl_rng, n):
        a = self._blk(philox_ref.P_TASK_A, j)
        b = self._blk(philox_ref.P_TASK_B, j)
        wl = philox_ref.uniform(philox_ref.d0(a), *wl_rng)
        data = philox_ref.uniform(philox_ref.d1(a), *data_rng)
        off = philox_ref.uniform(philox_ref.d0(b), *ddl_rng)
        dec = philox_ref.bounded(b[2], n)
        return wl, data, off, dec

    def explore_u(self, k, agent):
        return philox_ref.d0(self._blk(philox_ref.P_EXPL, agent))

    def explore_action(self, k, agent, n):
        blk = self._blk(philox_ref.P_EXPL, agent)
        blk2 = self._blk(philox_ref.P_EXPL_RES, agent)
        return philox_ref.bounded(blk[2], n), philox_ref.uniform(philox_ref.d0(blk2), 0.1, 1.0)

    def net(self, k, agent, bw_rng, lat_rng):
        blk = self._blk(philox_ref.P_NET, agent)
        return philox_ref.uniform(philox_ref.d0(blk), *bw_rng), philox_ref.uniform(philox_ref.d1(blk), *lat_rng)


# ----------------------------------------------------------------------------------------------
# structured tape writer
# ----------------------------------------------------------------------------------------------
def empty_tape(n_steps_plus_reset: int, n: int, a_slots: int = A_SLOTS) -> dict:
    s = n_steps_plus_reset
    return {
        "n_act": np.zeros(s, np.int32),
        "explore_u": np.full((s, n), np.nan),
        "explore_tgt": np.full((s, n), -1, np.int32),
        "explore_res": np.full((s, n), np.nan),
        "bw_u": np.full((s, n), np.nan),
        "lat": np.full((s, n), np.nan),
        "rate_u": np.full(s, np.nan),
        "dt_raw": np.full(s, np.nan),
        "n_arr": np.zeros(s, np.int32),
        "task_wl": np.zeros((s, a_slots)),
        "task_data": np.zeros((s, a_slots)),
        "task_ddl_off": np.zeros((s, a_slots)),
        "task_decider": np.zeros((s, a_slots), np.int32),
        "n_draws": np.zeros(s, np.int32),
    }


@dataclass
class StepOut:
    obs: np.ndarray            # [N, N+4] float32, zero rows for agents without a pending task
    order: list                # agent indices with a pending task, in arrival order
    reward: float              # global reward of the step (before the drop rule)
    reward_emitted: bool       # False when the new observation set is empty (MRRL.PY:454-455)
    done: bool
    metrics: dict | None
    latencies: list = field(default_factory=list)
    energies: list = field(default_factory=list)
    violations: list = field(default_factory=list)
    targets: list = field(default_factory=list)
    dt: float = 0.0
    rate: float = 0.0


class EdgeEnvOracle:
    """Array-API restatement; ``DictEnvOracle`` below adapts it to the reference's dict API."""

    def __init__(self, config=None, draws=None, tape_out=None):
        cfg = default_config() if config is None else config
        self.n = int(cfg["n_nodes"])
        # Scaled topology (build-defined, NOT in the reference): D decider agents over
        # S = n_nodes servers, home server of decider d = d mod S.  D == S (the default) is the reference.
        self.D = int(cfg.get("n_deciders", self.n))
        self.rate_fixed = cfg.get("task_arrival_rate", 10)
        self.rate_range = cfg.get("task_arrival_rate_range", (self.rate_fixed, self.rate_fixed))
        self.rate_mode = cfg.get("task_arrival_mode", "cyclic")
        self.rate_cycle = cfg.get("task_arrival_cycle_seconds", 20.0)
        self.horizon = int(cfg["episode_length"])
        self.cap = np.asarray(cfg["node_cpu_capacity"], dtype=np.float64)
        self.wl_rng = tuple(cfg["task_workload_range"])
        self.data_rng = tuple(cfg["task_data_range"])
        self.ddl_rng = tuple(cfg["task_deadline_range"])
        self.bw_rng = tuple(cfg["network_bw_range"])
        self.lat_rng = tuple(cfg["network_latency_range"])
        self.w = dict(cfg["reward_weights"])
        self.dt_min = cfg.get("min_time_step", 0.02)
        self.dt_max = cfg.get("max_time_step", 0.2)
        self.window0 = cfg.get("initial_observation_window", 0.1)
        self.sched = dict(cfg.get("exploration_schedule",
                                  {"initial_epsilon": 1.0, "final_epsilon": 0.05, "decay_steps": 1_000_000}))
        self.explore_off = bool(cfg.get("disable_env_exploration", False))
        self.draws = draws if draws is not None else LegacyDraws(None, np.random.mtrand._rand)
        self.tape_out = tape_out
        self.explore_counter = 0  # the reference keeps this on the class (MRRL.PY:122)
        self._zero()

    # -- state ---------------------------------------------------------------------------------
    def _zero(self):
        n = self.n
        self.load = np.zeros(n)
        self.queue = np.zeros(n)
        self.pending = {}           # agent -> (wl, data, deadline); insertion order = arrival order
        self.step_no = 0
        self.sim_time = 0.0
        self.elapsed = 0.0
        self.ep_lat, self.ep_en = [], []
        self.ep_viol = 0
        self.ep_done_tasks = 0
        self.ep_usage = []

    # -- MRRL.PY:220-235 -------------------------------------------------------------------------
    def _rate(self, t):
        lo, hi = self.rate_range
        if lo == hi:
            return lo
        if self.rate_mode == "cyclic":
            cyc = max(self.rate_cycle, 1e-6)
            phase = (t % cyc) / cyc
            if getattr(self.draws, "det_math", False):   # throughput mode: build-defined sine
                return lo + (0.5 + 0.5 * philox_ref.det_sin2pi(phase)) * (hi - lo)
            return lo + (0.5 + 0.5 * np.sin(2 * np.pi * phase)) * (hi - lo)
        if self.rate_mode == "random":
            u = self.draws.rate_u(lo, hi)
            self._rec("rate_u", u)
            self._count()
            return u
        return 0.5 * (lo + hi)

    def epsilon(self):
        # MRRL.PY:249-256
        if self.explore_off:
            return 0.0
        e0 = self.sched.get("initial_epsilon", 1.0)
        e1 = self.sched.get("final_epsilon", 0.05)
        span = max(1, self.sched.get("decay_steps", 1_000_000))
        frac = min(1.0, self.explore_counter / span)
        return e0 - (e0 - e1) * frac

    # -- tape recording --------------------------------------------------------------------------
    def _rec(self, key, val, slot=None):
        if self.tape_out is None:
            return
        s = self._tape_step
        if slot is None:
            self.tape_out[key][s] = val
        else:
            self.tape_out[key][s, slot] = val

    def _count(self, k=1):
        if self.tape_out is not None:
            self.tape_out["n_draws"][self._tape_step] += k

    # -- MRRL.PY:307-324 -------------------------------------------------------------------------
    def home(self, d):
        """Server a decider is attached to (its "local" placement): the identity in the reference."""
        return d % self.n

    def _obs_row(self, d, wl, data, deadline, t):
        ratio = self.load / self.cap
        h = self.home(d)
        head = [ratio[h], self.queue[h], data, wl, max(0, deadline - t)]
        rest = [ratio[j] for j in range(self.n) if j != h]
        return np.array(head + rest, dtype=np.float32)

    # -- MRRL.PY:340-383 -------------------------------------------------------------------------
    def _arrivals(self, t, advance):
        self.pending = {}
        obs = np.zeros((self.D, self.n + 4), np.float32)
        rate = self._rate(t)
        if advance:
            scale = 1.0 / max(rate, 1e-6)
            raw = self.draws.dt_raw(scale)
            self._rec("dt_raw", raw)
            self._count()
            window = float(np.clip(raw, self.dt_min, self.dt_max))
        else:
            window = self.window0
        lam = max(rate, 1e-6) * window
        n_new = self.draws.n_arrivals(lam)
        self._rec("n_arr", n_new)
        self._count()
        for j in range(n_new):
            wl, data, off, dec = self.draws.task(j, self.wl_rng, self.data_rng, self.ddl_rng, self.D)
            if self.tape_out is not None:
                if j >= self.tape_out["task_wl"].shape[1]:
                    raise AssertionError("tape arrival slots exhausted")
                self._rec("task_wl", wl, j)
                self._rec("task_data", data, j)
                self._rec("task_ddl_off", off, j)
                self._rec("task_decider", dec, j)
            self._count(4)
            deadline = t + off
            if dec in self.pending:
                self.queue[self.home(dec)] += 1
            else:
                self.pending[dec] = (wl, data, deadline)
                obs[dec] = self._obs_row(dec, wl, data, deadline, t)
        decay = window if advance else 0.0
        processed = self.cap * decay
        self.load -= processed
        self.load = np.maximum(0, self.load)
        if advance:
            self.queue -= 1
            self.queue = np.maximum(0, self.queue)
        denom = self.cap + 1e-9
        self.ep_usage.append(np.where(denom > 0, processed / denom, 0.0))
        return obs, (window if advance else 0.0), rate

    # -- MRRL.PY:188-212 -------------------------------------------------------------------------
    def reset(self, seed=None):
        if seed is not None:
            self.draws.reseed(seed)
        self._zero()
        self._tape_step = 0
        self.draws.begin_step(0)
        obs, _, _ = self._arrivals(self.sim_time, advance=False)
        return obs, list(self.pending.keys())

    # -- MRRL.PY:386-521 -------------------------------------------------------------------------
    def step(self, placement, resource, present=None):
        """placement[N] int, resource[N] float32; agents with a pending task act in arrival order.
        ``present`` optionally lists the agents whose action was supplied (dict API)."""
        self.step_no += 1
        self._tape_step = self.step_no
        self.draws.begin_step(self.step_no)
        t = self.sim_time
        lats, ens, viols, targets = [], [], [], []
        completed = 0
        k = 0
        for agent in list(self.pending.keys()):
            if present is not None and agent not in present:
                continue
            wl, data, deadline = self.pending[agent]
            target = int(placement[agent])
            res = np.float32(resource[agent])
            if not self.explore_off:
                eps = self.epsilon()
                u = self.draws.explore_u(k, agent)
                self._rec("explore_u", u, k)
                self._count()
                if not (u > eps):
                    target, r64 = self.draws.explore_action(k, agent, self.n)
                    self._rec("explore_tgt", target, k)
                    self._rec("explore_res", r64, k)
                    self._count(2)
                    res = np.array([r64], dtype=np.float32)[0]
            if target == self.home(agent):
                t_trans = 0.0
            else:
                bw_u, lat = self.draws.net(k, agent, self.bw_rng, self.lat_rng)
                self._rec("bw_u", bw_u, k)
                self._rec("lat", lat, k)
                self._count(2)
                bw = bw_u * 1e6 / 8
                t_trans = (data / bw) + lat
            t_queue = self.load[target] / self.cap[target]
            alloc = self.cap[target] * res          # float64 * float32 -> float64
            t_exec = wl / (alloc / 1e9)
            energy = wl * (alloc / 1e9)
            t_total = t_trans + t_queue + t_exec
            lats.append(t_total)
            ens.append(energy)
            if (t + t_total) > deadline:
                viols.append(1)
            else:
                viols.append(0)
                completed += 1
            self.load[target] += t_exec * (alloc / 1e9)
            self.queue[target] += 1
            targets.append(target)
            k += 1
        if self.tape_out is not None:
            self.tape_out["n_act"][self.step_no] = k

        obs, dt, rate = self._arrivals(t, advance=True)
        if dt > 0:
            self.sim_time += dt
            self.elapsed += dt

        if lats:
            cost = (self.w["alpha"] * np.mean(lats)) + (self.w["beta"] * np.mean(ens)) + (self.w["gamma"] * np.sum(viols))
            reward = -cost
        else:
            reward = 0.0
        done = self.step_no >= self.horizon
        self.ep_lat.extend(lats)
        self.ep_en.extend(ens)
        self.ep_viol += sum(viols)
        self.ep_done_tasks += completed
        if not self.explore_off:
            self.explore_counter += 1
        metrics = self.episode_metrics() if done else None
        return StepOut(obs=obs, order=list(self.pending.keys()), reward=float(reward),
                       reward_emitted=bool(self.pending), done=done, metrics=metrics,
                       latencies=lats, energies=ens, violations=viols, targets=targets, dt=dt, rate=float(rate))

    # -- MRRL.PY:470-504 -------------------------------------------------------------------------
    def episode_metrics(self):
        total = len(self.ep_lat) or 1
        if self.ep_lat:
            ms = np.array(self.ep_lat) * 1000
            avg_lat = float(np.mean(ms))
            p99 = float(np.percentile(ms, 99))
        else:
            avg_lat = p99 = 0.0
        avg_en = float(np.mean(self.ep_en)) if self.ep_en else float("nan")
        usage = np.mean(np.array(self.ep_usage), axis=0) if self.ep_usage else np.zeros(self.n)
        s1 = np.sum(usage)
        s2 = np.sum(usage ** 2)
        return {
            "avg_latency_ms": avg_lat,
            "p99_latency_ms": p99,
            "avg_energy_J": avg_en,
            "violation_rate": self.ep_viol / total,
            "throughput_tps": self.ep_done_tasks / max(self.elapsed, 1e-6),
            "load_balance_jain": float((s1 ** 2) / (self.n * s2 + 1e-9)),
        }


def convert_action(action, n):
    """``EdgeSimGym._convert_action`` for a continuous vector (MRRL.PY:283-305): (placement, float32 resource)."""
    vec = np.asarray(action, dtype=np.float32)
    if vec.ndim == 0:
        vec = vec.reshape(1)
    if vec.size < n + 1:
        padded = np.zeros(n + 1, dtype=np.float32)
        padded[:vec.size] = vec
        vec = padded
    logits = vec[:-1]
    placement = 0 if np.allclose(logits, 0) else int(np.argmax(logits)) % n
    norm = (float(vec[-1]) + 1.0) / 2.0
    value = 0.1 + 0.9 * np.clip(norm, 0.0, 1.0)
    return placement, np.array([value], dtype=np.float32)[0]


def greedy_action(load, cap):
    """The greedy baseline's action (MRRL.PY:806-812): (argmin(load/cap), float32(0.5))."""
    return int(np.argmin(np.asarray(load, np.float64) / np.asarray(cap, np.float64))), np.float32(0.5)


class DictEnvOracle:
    """Adapter with the reference's ``reset()/step(action_dict)`` dict API (MRRL.PY:188,386)."""

    def __init__(self, config=None, draws=None):
        self.core = EdgeEnvOracle(config, draws)
        self.n_nodes = self.core.n

    @property
    def disable_env_exploration(self):
        return self.core.explore_off

    @disable_env_exploration.setter
    def disable_env_exploration(self, v):
        self.core.explore_off = bool(v)

    def _obs_dict(self, obs, order):
        return {f"node_{a}": obs[a].copy() for a in order}

    def reset(self, *, seed=None, options=None):
        obs, order = self.core.reset(seed)
        return self._obs_dict(obs, order), {f"node_{a}": {} for a in order}

    def step(self, action_dict):
        n = self.core.n
        placement = np.zeros(n, np.int64)
        resource = np.zeros(n, np.float32)
        present = set()
        for name, act in action_dict.items():
            a = int(name.split("_")[1])
            present.add(a)
            placement[a] = int(act["placement"])
            resource[a] = np.float32(act["resource"][0])
        out = self.core.step(placement, resource, present)
        obs = self._obs_dict(out.obs, out.order)
        rewards = {k: out.reward for k in obs}
        infos = {k: {} for k in obs}
        if out.done:
            targets = list(obs.keys()) or ["__common__"]
            for k in targets:
                info = infos.setdefault(k, {})
                info["episode_metrics"] = dict(out.metrics)
                info["custom_metrics"] = dict(out.metrics)
                info.update(out.metrics)
        return obs, rewards, {"__all__": out.done}, {"__all__": False}, infos
