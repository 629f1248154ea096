"""Step time of em_step against the batch size and the workgroup size (tail / residency quantisation).
Usage: python tools/env_batch_sweep.py B[,B...] [waves[,waves...]]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.env import BatchedEdgeSimGym  # noqa: E402

Bs = [int(x) for x in sys.argv[1].split(",")]
waves = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
dev = torch.device("cuda:0")
for B in Bs:
    for w in waves:
        cfg = dict(ENV_CONFIG, disable_env_exploration=True)
        env = BatchedEdgeSimGym(cfg, n_envs=B, device=dev, seed=0)
        if w:
            env.set_workgroup_waves(w)
        n = env.n_nodes
        gen = torch.Generator(device=dev).manual_seed(1234)
        pool = [(torch.randint(0, n, (B, n), device=dev, generator=gen, dtype=torch.int32),
                 (torch.rand((B, n), device=dev, generator=gen) * 0.9 + 0.1).contiguous()) for _ in range(8)]
        env.reset()
        L, h, out = env._L, env._h, C.byref(env._out)
        s = torch.cuda.current_stream(dev).cuda_stream
        def run(k):
            for i in range(k):
                p, r = pool[i % 8]
                assert L.em_step(h, p.data_ptr(), r.data_ptr(), None, out, s) == 0
        run(300)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(600); e1.record()
        torch.cuda.synchronize()
        v = env.kernel_variant()
        print(f"B={B:6d} waves={v['workgroup_waves']} lanes={v['lanes_per_env']} packed={v['packed']}: {e0.elapsed_time(e1) / 600 * 1e3:7.2f} us/step", flush=True)
        env.close()
