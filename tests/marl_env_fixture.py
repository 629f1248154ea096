"""Helpers shared by the EdgeMARLEnv tests: load a golden fixture (recorded from the reference by
oracle/make_golden_marl.py), replay it through the restatement, and build the structured tape the kernel consumes."""
import json
import os

import numpy as np

from oracle import edge_marl_oracle as mo

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = ("cyclic_discrete", "random_hybrid", "fixed_small", "cyclic_wide")


def load(name):
    z = np.load(os.path.join(GOLDEN, f"marlenv_{name}.npz"))
    args = json.loads(bytes(z["config_json"]).decode())
    names = [str(s) for s in z["draw_names"]]
    log = []
    for k, a, v in zip(z["draw_kind"], z["draw_args"], z["draw_value"]):
        log.append((names[k], tuple(float(x) for x in a if not np.isnan(x)), float(v)))
    return args, z, log


def replay(name):
    """Runs the restatement on the reference's recorded draws; returns (args, fixture, per-step outputs, tape)."""
    args, z, log = load(name)
    steps = len(z["reward"])
    n = int(args["n_nodes"])
    tape = mo.empty_tape(steps + 1, n)
    draws = mo.FlatReplay(log)
    env = mo.EdgeMARLOracle.__new__(mo.EdgeMARLOracle)
    # the recorded episode starts at the explicit reset(): build without the constructor's own reset draws
    env_args = dict(args)
    mo.EdgeMARLOracle.__init__(env, env_args, draws=_Quiet(), tape_out=None)
    env.draws, env.tape_out = draws, tape
    env.reset()
    out = {"obs": [env.obs()], "reward": [], "terminated": [], "info": []}
    for t in range(steps):
        r, term, info = env.step(z["actions"][t])
        out["reward"].append(r); out["terminated"].append(term)
        out["info"].append([info[k] for k in ("avg_latency_ms", "avg_energy_J", "deadline_violation_rate", "throughput_tps", "load_balance_jain")])
        out["obs"].append(env.obs())
    assert draws.pos == len(log), "restatement consumed a different number of draws than the reference"
    out = {k: np.array(v) for k, v in out.items()}
    out["summary"] = env.episode_summary()
    return args, z, out, tape


class _Quiet:
    """Draw source for the constructor's implicit reset (values are discarded)."""

    def begin_step(self, s):
        pass

    def task(self, agent, a, b, c):
        return 1.0, 1.0, 0.1
