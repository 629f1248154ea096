"""IPPO loop timing on the scaled topology (64 servers x 256 deciders): rollout + per-agent GAE + grouped PPO update."""
import argparse
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from marl_amd.config import SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.ippo import IPPOTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--no-overlap", action="store_true")
    ap.add_argument("--horizon", type=int, default=1000)
    ap.add_argument("--shared", action="store_true")
    ap.add_argument("--fwd-split", type=int, default=0)
    ap.add_argument("--bwd-split", type=int, default=0)
    ap.add_argument("--cc-split", type=int, default=0)
    ap.add_argument("--algorithm", default="ippo", choices=["ippo", "shared", "mappo"])
    a = ap.parse_args()
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["episode_length"] = a.horizon
    tr = IPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=a.envs, algorithm="shared" if a.shared else a.algorithm, verbose=False)
    tr.overlap_value = not a.no_overlap
    if a.fwd_split:
        tr.algo.fwd_split = a.fwd_split
    if a.cc_split:
        tr.algo.cc_split = a.cc_split
        tr.algo._cc_slab = None
    if a.bwd_split:
        tr.algo.split = a.bwd_split
        tr.algo._slab_a = tr.algo._slab_c = None
    for it in range(a.iters + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.train_episode(explore=True, record_buffer=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        E = tr.algo.buffer.total()
        out = tr.algo.update()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tag = "warmup" if it == 0 else f"iter{it}"
        print(f"{tag}: envs={a.envs} T={a.horizon} entries={E} rollout={t1-t0:.3f}s update={t2-t1:.3f}s total={t2-t0:.3f}s "
              f"env-steps/s={a.envs*a.horizon/(t2-t0):.3e} agent-steps/s={a.envs*a.horizon*256/(t2-t0):.3e} "
              f"samples/s={E/(t2-t0):.3e} losses={out}", flush=True)


if __name__ == "__main__":
    main()
