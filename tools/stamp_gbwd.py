"""Cycle stamps of gmlp_bwd_kernel (library built with -DEM_STAMP_GBWD): cycles per phase of one activation wavefront, per tile."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.config import SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.ippo import IPPOTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
tr = IPPOTrainer(dict(SCALED_ENV_CONFIG, episode_length=100), seed=0, device="cuda:0", n_envs=B, algorithm="ippo", verbose=False)
L = _lib.lib()
tr.train_episode(explore=True, record_buffer=True)
torch.cuda.synchronize()
assert L.em_debug_gbwd_stamps(None, 1) == 0
tr.algo.update()
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
assert L.em_debug_gbwd_stamps(st, 0) == 0
names = ["", "top (hand-over wait, x fragment, loss inputs)", "forward", "loss", "dW3 operands out", "dz2, dz1", "dW3 / wait",
         "tr(dZ2), tr(H1), dW2", "dZ1 out / dW1"]
for which, label in ((0, "critic"), (1, "actor")):
    n = st[which * 16]
    if not n:
        continue
    tot = sum(st[which * 16 + i] for i in range(1, 9))
    print(f"{label}: {n} tiles, {tot / n:.0f} cycles per tile")
    for i in range(1, 9):
        v = st[which * 16 + i] / n
        print(f"   {names[i]:45s} {v:8.0f}  {100 * v * n / tot:5.1f} %")
