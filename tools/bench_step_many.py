"""Ad-hoc timing of em_step_many (k steps per launch) against single-step launches."""
import sys
import time

import torch

sys.path.insert(0, ".")
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.env import BatchedEdgeSimGym  # noqa: E402

cfg = dict(ENV_CONFIG, disable_env_exploration=True)
for B in (4096, 16384, 65536):
    n, K = 10, 250
    env = BatchedEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=1, latency_log_capacity=0)
    pl = torch.randint(0, n, (K, B, n), device="cuda", dtype=torch.int32)
    rs = (torch.rand((K, B, n), device="cuda") * 0.9 + 0.1).contiguous()
    for mode in ("single", "many", "many+obs_history", "greedy_many"):
        env.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(4):
            if rep == 1:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            if mode == "single":
                for t in range(K):
                    env.step(pl[t], rs[t])
            elif mode == "many":
                env.step_many(K, placement=pl, resource=rs)
            elif mode == "many+obs_history":
                env.step_many(K, placement=pl, resource=rs, obs_history=True)
            else:
                env.step_many(K, greedy_share=0.5)
            if rep < 3 and False:
                pass
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (3 * K)
        print(f"B={B} {mode:18s}: {dt*1e6:7.2f} us/step  {B*n/dt/1e9:6.2f} G agent-steps/s", flush=True)
    env.close()
