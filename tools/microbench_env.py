"""Ad-hoc timing of em_step for several batch sizes and kernel layouts (not the graded bench).
lanes: 16 = power-of-two lane groups, 1 = packed groups."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.env import BatchedEdgeSimGym  # noqa: E402

batches = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [4096, 16384, 32768, 65536, 262144]
modes = [bool(int(x)) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [False, True]
for explore in modes:
    cfg = dict(ENV_CONFIG, disable_env_exploration=not explore)
    for B in batches:
        for lanes in (16, 1):
            env = BatchedEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=1)
            env.set_lanes_per_env(lanes)
            n = env.n_nodes
            pool = [(torch.randint(0, n, (B, n), device="cuda", dtype=torch.int32), torch.rand((B, n), device="cuda") * 0.9 + 0.1)
                    for _ in range(8)]
            env.reset()
            L, h, out = env._L, env._h, C.byref(env._out)
            stream = torch.cuda.current_stream().cuda_stream
            for i in range(50):
                env.step(*pool[i % 8])
            torch.cuda.synchronize()
            K = 400
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ptrs = [(p.data_ptr(), r.data_ptr()) for p, r in pool]
            t0 = time.perf_counter()
            e0.record()
            for i in range(K):
                pp, rr = ptrs[i % 8]
                L.em_step(h, pp, rr, None, out, stream)
            e1.record()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            ms = e0.elapsed_time(e1) / K
            bytes_step = B * (2 * env.record_bytes + 8 * n + 4 * n * (n + 4) + 8 + 8 + 8 + 1 + 1 + n)
            print(f"explore={int(explore)} B={B} lanes={lanes}: {ms*1e3:.2f} us/step (host wall {wall/K*1e6:.2f} us)  "
                  f"{B*n/ms/1e6:.2f} G agent-steps/s  {bytes_step/ms/1e9:.3f} TB/s", flush=True)
            env.close()
