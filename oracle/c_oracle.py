"""ctypes wrapper around oracle/_build/libedge_oracle.so (built from oracle/edge_oracle.c).

TEST INFRASTRUCTURE ONLY - see the header of edge_oracle.c.  ``build()`` compiles the library with
gcc if it is missing (the GPU box receives the prebuilt file with the repository snapshot).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libedge_oracle.so")
EO_MAX_N = 64


class EoConfig(C.Structure):
    _fields_ = [
        ("n", C.c_int32), ("horizon", C.c_int32), ("rate_mode", C.c_int32), ("explore_off", C.c_int32),
        ("cap", C.c_double * EO_MAX_N),
        ("rate_lo", C.c_double), ("rate_hi", C.c_double), ("cycle", C.c_double),
        ("wl_lo", C.c_double), ("wl_hi", C.c_double), ("data_lo", C.c_double), ("data_hi", C.c_double),
        ("ddl_lo", C.c_double), ("ddl_hi", C.c_double),
        ("bw_lo", C.c_double), ("bw_hi", C.c_double), ("lat_lo", C.c_double), ("lat_hi", C.c_double),
        ("w_lat", C.c_double), ("w_en", C.c_double), ("w_viol", C.c_double),
        ("dt_min", C.c_double), ("dt_max", C.c_double), ("window0", C.c_double),
        ("eps0", C.c_double), ("eps1", C.c_double),
        ("decay_steps", C.c_int64),
    ]


class EoTape(C.Structure):
    _fields_ = [
        ("explore_u", C.c_void_p), ("explore_res", C.c_void_p), ("bw_u", C.c_void_p), ("lat", C.c_void_p),
        ("explore_tgt", C.c_void_p),
        ("rate_u", C.c_void_p), ("dt_raw", C.c_void_p),
        ("n_arr", C.c_void_p),
        ("task_wl", C.c_void_p), ("task_data", C.c_void_p), ("task_ddl_off", C.c_void_p),
        ("task_decider", C.c_void_p),
        ("a_slots", C.c_int32),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "edge_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.eo_create.restype = C.c_void_p
        L.eo_create.argtypes = [C.POINTER(EoConfig), C.c_int64, C.c_uint64, C.c_uint64]
        L.eo_destroy.argtypes = [C.c_void_p]
        L.eo_set_explore.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
        L.eo_get_explore_counter.restype = C.c_int64
        L.eo_get_explore_counter.argtypes = [C.c_void_p]
        L.eo_set_seed.argtypes = [C.c_void_p, C.c_uint64]
        L.eo_next_episode.argtypes = [C.c_void_p]
        L.eo_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.eo_step.argtypes = [C.c_void_p] + [C.c_void_p] * 11
        L.eo_metrics.argtypes = [C.c_void_p, C.c_void_p]
        L.eo_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.eo_gae.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_double,
                             C.c_void_p, C.c_void_p]
        L.np_pairwise_sum.restype = C.c_double
        L.np_pairwise_sum.argtypes = [C.c_void_p, C.c_int64]
        L.np_percentile_linear.restype = C.c_double
        L.np_percentile_linear.argtypes = [C.c_void_p, C.c_int64, C.c_double]
        for f in ("eo_det_log", "eo_det_exp", "eo_det_sin2pi"):
            getattr(L, f).restype = C.c_double
            getattr(L, f).argtypes = [C.c_double]
        L.eo_philox.argtypes = [C.c_uint32] * 4 + [C.c_uint64, C.c_void_p]
        _lib = L
    return _lib


_MODES = {"cyclic": 1, "random": 2}


def make_config(cfg: dict) -> EoConfig:
    c = EoConfig()
    n = int(cfg["n_nodes"])
    c.n = n
    c.horizon = int(cfg["episode_length"])
    lo, hi = cfg.get("task_arrival_rate_range", (cfg.get("task_arrival_rate", 10),) * 2)
    c.rate_lo, c.rate_hi = float(lo), float(hi)
    c.rate_mode = _MODES.get(cfg.get("task_arrival_mode", "cyclic"), 0)
    c.cycle = float(cfg.get("task_arrival_cycle_seconds", 20.0))
    c.explore_off = int(bool(cfg.get("disable_env_exploration", False)))
    caps = np.asarray(cfg["node_cpu_capacity"], np.float64)
    for i in range(n):
        c.cap[i] = float(caps[i])
    c.wl_lo, c.wl_hi = map(float, cfg["task_workload_range"])
    c.data_lo, c.data_hi = map(float, cfg["task_data_range"])
    c.ddl_lo, c.ddl_hi = map(float, cfg["task_deadline_range"])
    c.bw_lo, c.bw_hi = map(float, cfg["network_bw_range"])
    c.lat_lo, c.lat_hi = map(float, cfg["network_latency_range"])
    w = cfg["reward_weights"]
    c.w_lat, c.w_en, c.w_viol = float(w["alpha"]), float(w["beta"]), float(w["gamma"])
    c.dt_min = float(cfg.get("min_time_step", 0.02))
    c.dt_max = float(cfg.get("max_time_step", 0.2))
    c.window0 = float(cfg.get("initial_observation_window", 0.1))
    sch = cfg.get("exploration_schedule", {})
    c.eps0 = float(sch.get("initial_epsilon", 1.0))
    c.eps1 = float(sch.get("final_epsilon", 0.05))
    c.decay_steps = int(sch.get("decay_steps", 1_000_000))
    return c


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class TapeStep:
    """Holds the numpy arrays of one tape step for B envs and the C struct pointing at them."""

    KEYS_F64_N = ("explore_u", "explore_res", "bw_u", "lat")

    def __init__(self, arrays: dict, B: int, n: int):
        a = {k: np.ascontiguousarray(v) for k, v in arrays.items()}
        self.arrays = a
        t = EoTape()
        for k in ("explore_u", "explore_res", "bw_u", "lat", "rate_u", "dt_raw", "task_wl", "task_data", "task_ddl_off"):
            assert a[k].dtype == np.float64
            setattr(t, k, a[k].ctypes.data)
        for k in ("explore_tgt", "n_arr", "task_decider"):
            assert a[k].dtype == np.int32
            setattr(t, k, a[k].ctypes.data)
        t.a_slots = a["task_wl"].shape[-1]
        self.struct = t


def tape_step_from_golden(tapes: list, s: int) -> dict:
    """Stacks step ``s`` of several single-env structured tapes into [B, ...] arrays."""
    keys = ("explore_u", "explore_res", "bw_u", "lat", "explore_tgt", "rate_u", "dt_raw", "n_arr",
            "task_wl", "task_data", "task_ddl_off", "task_decider")
    return {k: np.ascontiguousarray(np.stack([t[k][s] for t in tapes])) for k in keys}


class COracleEnv:
    """Batched C restatement.  ``tape=None`` in reset/step selects the Philox stream."""

    def __init__(self, cfg: dict, B: int, seed: int = 0, env_offset: int = 0):
        self.L = lib()
        self.cfg = cfg
        self.n = int(cfg["n_nodes"])
        self.B = int(B)
        self._c = make_config(cfg)
        self.h = self.L.eo_create(C.byref(self._c), self.B, int(seed), int(env_offset))
        if not self.h:
            raise RuntimeError("eo_create failed")
        n, B = self.n, self.B
        self.obs = np.zeros((B, n, n + 4), np.float32)
        self.mask = np.zeros((B, 2), np.uint32)
        self.rank = np.zeros((B, n), np.int8)
        self.reward = np.zeros(B)
        self.emitted = np.zeros(B, np.uint8)
        self.done = np.zeros(B, np.uint8)
        self.step_lat = np.zeros((B, n))
        self.n_draws = np.zeros(B, np.int32)

    def __del__(self):
        try:
            self.L.eo_destroy(self.h)
        except Exception:
            pass

    def set_explore(self, off: bool, counter: int):
        self.L.eo_set_explore(self.h, int(off), int(counter))

    @property
    def explore_counter(self):
        return int(self.L.eo_get_explore_counter(self.h))

    def set_seed(self, seed):
        self.L.eo_set_seed(self.h, int(seed))

    def next_episode(self):
        self.L.eo_next_episode(self.h)

    def reset(self, tape: TapeStep | None = None):
        self.L.eo_reset(self.h, C.byref(tape.struct) if tape else None, _p(self.obs), _p(self.mask), _p(self.rank))
        return self.obs, self.mask[:, 0].astype(np.uint64) | (self.mask[:, 1].astype(np.uint64) << np.uint64(32)), self.rank

    def step(self, placement, resource, tape: TapeStep | None = None):
        pl = np.ascontiguousarray(placement, np.int32).reshape(self.B, self.n)
        rs = np.ascontiguousarray(resource, np.float32).reshape(self.B, self.n)
        self.L.eo_step(self.h, _p(pl), _p(rs), C.byref(tape.struct) if tape else None, _p(self.obs), _p(self.mask),
                       _p(self.rank), _p(self.reward), _p(self.emitted), _p(self.done), _p(self.step_lat),
                       _p(self.n_draws))
        return self.obs, self.reward, self.emitted, self.done

    def metrics(self):
        out = np.zeros((self.B, 6))
        self.L.eo_metrics(self.h, _p(out))
        return out

    def state(self):
        B, n = self.B, self.n
        st = {"load": np.zeros((B, n)), "queue": np.zeros((B, n)), "sim_time": np.zeros(B),
              "step": np.zeros(B, np.int32), "counts": np.zeros((B, 3), np.int64), "elapsed": np.zeros(B)}
        self.L.eo_get_state(self.h, _p(st["load"]), _p(st["queue"]), _p(st["sim_time"]), _p(st["step"]),
                            _p(st["counts"]), _p(st["elapsed"]))
        return st


def gae(rewards, values, dones, next_value, gamma, lam):
    r = np.ascontiguousarray(rewards, np.float64)
    v = np.ascontiguousarray(values, np.float64)
    d = np.ascontiguousarray(dones, np.uint8)
    adv = np.zeros_like(r)
    ret = np.zeros_like(r)
    lib().eo_gae(_p(r), _p(v), _p(d), len(r), float(next_value), float(gamma), float(lam), _p(adv), _p(ret))
    return adv, ret
