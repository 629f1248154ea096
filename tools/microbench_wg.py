"""Ad-hoc: em_step launch time against wavefronts per workgroup and lane packing at small batches."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.env import BatchedEdgeSimGym  # noqa: E402

cfg = dict(ENV_CONFIG)
cfg["disable_env_exploration"] = True
for B in (4096, 16384):
    for lanes in (16, 1):
        for waves in (2, 3, 4, 5, 6, 8):
            env = BatchedEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=1, latency_log_capacity=0)
            env.set_lanes_per_env(lanes)
            env.set_workgroup_waves(waves)
            n = env.n_nodes
            pool = [(torch.randint(0, n, (B, n), device="cuda", dtype=torch.int32), torch.rand((B, n), device="cuda") * 0.9 + 0.1)
                    for _ in range(8)]
            env.reset()
            L, h, out = env._L, env._h, C.byref(env._out)
            stream = torch.cuda.current_stream().cuda_stream
            ptrs = [(p.data_ptr(), r.data_ptr()) for p, r in pool]
            for i in range(100):
                L.em_step(h, *ptrs[i % 8], None, out, stream)
            torch.cuda.synchronize()
            K = 800
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(K):
                L.em_step(h, *ptrs[i % 8], None, out, stream)
            e1.record()
            torch.cuda.synchronize()
            print(f"B={B} lanes={lanes} wg_waves={waves}: {e0.elapsed_time(e1) / K * 1e3:.2f} us/step", flush=True)
            env.close()
