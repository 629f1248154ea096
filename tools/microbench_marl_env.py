"""Times emm_step (EdgeMARLEnv) at several batch sizes: us per launch, agent-steps/s, algorithmic GB/s."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for B in [int(x) for x in (sys.argv[1:] or ["4096", "65536", "262144"])]:
        for lanes in (1, 16):
            if lanes == 16 and B > 65536:
                continue
            r = bench.marl_env_leg(dev, 0, 1, B=B, steps=300, lanes=lanes)
            print(f"B={B:7d} lanes/env={lanes:2d}  {r['avg_launch_us']:8.2f} us/step  {r['value'] / 1e9:7.2f} G agent-steps/s  "
                  f"{r['roofline']['achieved']:7.1f} GB/s ({r['roofline']['frac']:.3f} of HBM peak)", flush=True)
