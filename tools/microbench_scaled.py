"""Ad-hoc timing of ems_step (scaled topology, 64 servers x 256 deciders) for several batch sizes (not the graded bench)."""
import ctypes as C
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from marl_amd.config import SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.env_scaled import PENDING_CAP, ScaledEdgeSimGym  # noqa: E402


def algorithmic_bytes(S, k_old, k_new):
    """per env-step: fixed record part read+written, pending entries read/written, actions of acting slots, compact outputs"""
    return 2 * (12 * S + 64) + 28 * (k_old + k_new) + 8 * k_old + k_new * (4 * (S + 4) + 4) + 4 + 8 + 8 + 1 + 1


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [4096, 65536]
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["disable_env_exploration"] = True
    S = cfg["n_nodes"]
    for B in sizes:
        env = ScaledEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=1)
        pool = [(torch.randint(0, S, (B, PENDING_CAP), device="cuda", dtype=torch.int32),
                 torch.rand((B, PENDING_CAP), device="cuda") * 0.9 + 0.1) for _ in range(8)]
        env.reset()
        for i in range(30):
            env.step(*pool[i % 8], by_slot=True)
        torch.cuda.synchronize()
        K = 300
        L, h, out = env._L, env._h, C.byref(env._out)
        stream = torch.cuda.current_stream().cuda_stream
        ptrs = [(p.data_ptr(), r.data_ptr()) for p, r in pool]
        ksum = torch.zeros((), dtype=torch.int64, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(K):
            pp, rr = ptrs[i % 8]
            L.ems_step(h, pp, rr, 1, out, stream)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / K
        for i in range(20):   # mean pending deciders per step (outside the timed region)
            env.step(*pool[i % 8], by_slot=True)
            ksum += env.n_pending.sum()
        kbar = ksum.item() / (20 * B)
        by = algorithmic_bytes(S, kbar, kbar) * B
        print(f"B={B}: {ms*1e3:.2f} us/step  {B/ms/1e6:.3f} G env-steps/s  {B*cfg['n_deciders']/ms/1e6:.1f} G agent-steps/s  "
              f"mean pending {kbar:.2f}  {by/ms/1e6:.1f} GB/s algorithmic ({by/B:.0f} B/env-step)", flush=True)
        env.close()


if __name__ == "__main__":
    main()
