"""Cycle stamps of scaled_env4_kernel (library built with -DEMS_STAMP): one wavefront's path through a step.
   python tools/stamp_scaled.py [n_envs]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd.config import SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.env_scaled import PENDING_CAP, ScaledEdgeSimGym  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = dict(SCALED_ENV_CONFIG, disable_env_exploration=True)
S = cfg["n_nodes"]
env = ScaledEdgeSimGym(cfg, n_envs=B, device="cuda:0", seed=1)
pl = torch.randint(0, S, (B, PENDING_CAP), device="cuda", dtype=torch.int32)
rs = torch.rand((B, PENDING_CAP), device="cuda") * 0.9 + 0.1
env.reset()
names = ["entry", "capacity tables + barrier", "loads issued", "Philox: arrivals", "slow test, LDS stage, action fix-up",
         "Philox: network draws", "placement terms", "ordered chain", "reward, logs", "ratios, duplicate arrivals", "rows, pending list out",
         "decay, outputs"]
acc = [0.0] * 12
n = 0
for i in range(60):
    env.step(pl, rs, by_slot=True)
    torch.cuda.synchronize()
    if i < 10:
        continue
    st = (C.c_ulonglong * 16)()
    assert env._L.ems_debug_stamps(st) == 0
    for j in range(1, 12):
        acc[j] += st[j] - st[j - 1]
    n += 1
tot = sum(acc)
print(f"B={B}: {tot / n:.0f} cycles per step (wavefront 0 of workgroup 7)")
for j in range(1, 12):
    print(f"   {names[j]:40s} {acc[j] / n:8.0f}  {100 * acc[j] / tot:5.1f} %")
