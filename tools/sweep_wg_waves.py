"""Headline step (4,096 environments) for every workgroup size the step kernel accepts: us per launch by HIP events.
    python tools/sweep_wg_waves.py [n_envs]"""
import ctypes as C
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.env import BatchedEdgeSimGym  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = dict(ENV_CONFIG)
cfg["disable_env_exploration"] = True
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1)
for waves in (0, 2, 3, 4, 5, 6, 7, 8, 9):
    env = BatchedEdgeSimGym(cfg, n_envs=B, device=dev, seed=0)
    try:
        if waves:
            env.set_workgroup_waves(waves)
    except RuntimeError as e:
        print(waves, "rejected:", str(e)[:60])
        continue
    n = env.n_nodes
    pl = torch.randint(0, n, (B, n), device=dev, generator=gen, dtype=torch.int32)
    rs = (torch.rand((B, n), device=dev, generator=gen) * 0.9 + 0.1).contiguous()
    L, h, out = env._L, env._h, C.byref(env._out)
    s = torch.cuda.current_stream(dev).cuda_stream
    args = (h, C.c_void_p(pl.data_ptr()), C.c_void_p(rs.data_ptr()), None, out, C.c_void_p(s))
    env.reset()
    for _ in range(300):
        L.em_step(*args)
    env.reset()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(900):
        L.em_step(*args)
    e1.record()
    torch.cuda.synchronize()
    print(f"waves {waves or 'default'}: {env.kernel_variant()}  {e0.elapsed_time(e1) / 900 * 1e3:.3f} us per launch")
    env.close()
