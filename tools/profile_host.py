"""Host-side cost of one rollout step: cProfile of train_episode at a batch small enough that the GPU is never the limit."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "mappo"
    if which == "mappo":
        from marl_amd.config import ENV_CONFIG
        from marl_amd.trainer import MAPPOTrainer
        tr = MAPPOTrainer(dict(ENV_CONFIG), seed=0, device="cuda:0", n_envs=256, verbose=False)
    else:
        from marl_amd.ippo import IPPOTrainer
        tr = IPPOTrainer(None, seed=0, device="cuda:0", n_envs=256, verbose=False, algorithm=which)
    tr.train_episode(explore=True, record_buffer=True)
    (tr.mappo if which == "mappo" else tr.algo).update()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    tr.train_episode(explore=True, record_buffer=True)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
