"""Test infrastructure (bench.py's cpu_baseline leg only): one worker of the "whole host" CPU figure of SURVEY.md 8d —
the Python/numpy restatement of EdgeSimGym.step (oracle/edge_env_oracle.py, one environment, random actions, exploration
off) stepped for a fixed time in its own process.  Prints one JSON line {"steps": ..., "seconds": ...}.

    python -m oracle.cpu_timing SECONDS SEED
"""
from __future__ import annotations

import json
import sys
import time

import numpy as np

from oracle import edge_env_oracle as eo


def run(seconds: float, seed: int) -> dict:
    cfg = eo.default_config()
    cfg["disable_env_exploration"] = True
    env = eo.EdgeEnvOracle(cfg, eo.LegacyDraws(seed))
    n = env.n
    rs = np.random.RandomState(1000 + seed)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.reset()
        for _ in range(cfg["episode_length"]):
            env.step(rs.randint(0, n, size=n), rs.uniform(0.1, 1.0, size=n).astype(np.float32))
        steps += cfg["episode_length"]
    return {"steps": steps, "seconds": time.perf_counter() - t0, "n": n}


if __name__ == "__main__":
    print(json.dumps(run(float(sys.argv[1]), int(sys.argv[2]))), flush=True)
