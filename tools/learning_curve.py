"""Prints the training-reward curve of MAPPO (default topology) and IPPO (small scaled topology) for a fixed seed:
the exploration run behind tests/test_learning_gpu.py's thresholds.  python tools/learning_curve.py [iters]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.config import ENV_CONFIG, SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.ippo import IPPOTrainer  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120


def curve(tr, algo, lr=None):
    out = []
    for _ in range(iters):
        st = tr.train_episode(explore=True, record_buffer=True)
        algo.update()
        out.append(st["episode_reward"])
    return np.array(out)


for lr in (3e-4, 3e-3):
    for seed in (0, 1):
        cfg = dict(ENV_CONFIG, episode_length=200, disable_env_exploration=True)
        t0 = time.time()
        tr = MAPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=256, verbose=False)
        tr.mappo.actor_optimizer.param_groups[0]["lr"] = lr
        tr.mappo.critic_optimizer.param_groups[0]["lr"] = lr
        c = curve(tr, tr.mappo)
        print(f"MAPPO lr={lr} seed={seed} first10={c[:10].mean():.2f} last10={c[-10:].mean():.2f} std_first={c[:10].std():.2f} "
              f"curve={np.round(c[::10], 1).tolist()} {time.time() - t0:.1f}s", flush=True)
        tr.env.close()

for lr in (3e-4, 3e-3):
    for seed in (0, 1):
        cfg = dict(SCALED_ENV_CONFIG, n_nodes=8, n_deciders=24, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9, episode_length=200,
                   disable_env_exploration=True, task_arrival_rate=20, task_arrival_rate_range=(10, 30))
        t0 = time.time()
        tr = IPPOTrainer(cfg, seed=seed, device="cuda:0", n_envs=256, verbose=False)
        tr.algo.lr = lr
        c = curve(tr, tr.algo)
        print(f"IPPO lr={lr} seed={seed} first10={c[:10].mean():.2f} last10={c[-10:].mean():.2f} std_first={c[:10].std():.2f} "
              f"curve={np.round(c[::10], 1).tolist()} {time.time() - t0:.1f}s", flush=True)
        tr.env.close()
