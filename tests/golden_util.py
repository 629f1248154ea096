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


ACTOR_TERM_SCALE = 0.79      # E|N(0,1)|: mean magnitude of one PPO surrogate term ratio * normalised advantage at ratio ~ 1


def assert_losses_close(got, want, label="", rel=1e-5):
    """north_star's bar for losses: 1e-5 relative in fp32.  ``got``/``want`` = (actor_loss, critic_loss, entropy).
    Critic loss and entropy are means of same-signed terms and are held relative to themselves.  The actor loss is a mean of
    SIGNED terms of magnitude ~0.8 (normalised advantages x ratio) that cancels to ~0 (it is exactly -mean(adv_norm) = 0 up
    to rounding in the first epoch), so 1e-5 of its own value would be below the float32 rounding of its summands; it is
    held to 1e-5 of the mean |summand| instead.  The measured errors are printed (pytest -s)."""
    import numpy as np
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.abs(want).copy()
    scale[0] = max(scale[0], ACTOR_TERM_SCALE)
    err = np.abs(got - want) / scale
    print(f"{label} loss errors relative to scale (actor, critic, entropy): " + ", ".join(f"{e:.2e}" for e in err))
    assert (err <= rel).all(), f"{label}: losses {got} vs {want}: scaled errors {err} exceed {rel}"
