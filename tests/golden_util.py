"""Helpers to load the golden fixtures written by oracle/make_golden.py (data only)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
METRIC_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "violation_rate", "throughput_tps",
               "load_balance_jain")


def env_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN, "env_*.npz")))


def load_env_fixture(path):
    z = np.load(path, allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["tape"] = {k[5:]: g.pop(k) for k in list(g) if k.startswith("tape_")}
    g["name"] = os.path.basename(path)
    return g


def config_for(g, base):
    cfg = dict(base)
    cfg["n_nodes"] = int(g["n_nodes"])
    cfg["episode_length"] = int(g["episode_length"])
    cfg["disable_env_exploration"] = bool(g["disable_env_exploration"])
    cfg["task_arrival_mode"] = str(g["arrival_mode"])
    lo, hi = g["rate_range"]
    cfg["task_arrival_rate_range"] = (float(lo), float(hi))
    sch = dict(cfg["exploration_schedule"])
    sch["decay_steps"] = int(g["decay_steps"])
    cfg["exploration_schedule"] = sch
    return cfg
