"""Rollout / update wall time of the 65,536-environment MAPPO loop (three iterations).  Usage: python tools/rollout_time.py [n_envs]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tr = MAPPOTrainer(dict(ENV_CONFIG), seed=0, device="cuda:0", n_envs=n_envs, verbose=False)
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_episode(explore=True, record_buffer=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    e, s = tr.mappo.buffer.totals()
    tr.mappo.update()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"rollout {(t1 - t0) * 1e3:.1f} ms update {(t2 - t1) * 1e3:.1f} ms ({e} entries)", flush=True)
