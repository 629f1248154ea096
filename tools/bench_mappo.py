"""Full MAPPO loop timing (rollout + GAE + PPO update) on one GPU: per-phase wall time and samples/s."""
import argparse
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--actor-per-cu", type=int, default=0, help="workgroups per CU of the actor backward (0: library default)")
    a = ap.parse_args()
    cfg = dict(ENV_CONFIG)
    cfg["episode_length"] = a.horizon
    tr = MAPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=a.envs, verbose=False)
    tr.overlap_value = not a.no_overlap
    if a.actor_per_cu:
        tr.mappo.actor_wg_per_cu = a.actor_per_cu
    for it in range(a.iters + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_episode(explore=True, record_buffer=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        E, S = tr.mappo.buffer.totals()
        tr.mappo.compute_gae()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out = tr.mappo.update()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        tag = "warmup" if it == 0 else f"iter{it}"
        print(f"{tag}: envs={a.envs} T={a.horizon} entries={E} steps={S} rollout={t1-t0:.3f}s gae={t2-t1:.4f}s "
              f"update(incl gae)={t3-t2:.3f}s total={t3-t0:.3f}s  env-steps/s={a.envs*a.horizon/(t3-t0):.3e} "
              f"samples/s={E/(t3-t0):.3e} losses={out}", flush=True)


if __name__ == "__main__":
    main()
