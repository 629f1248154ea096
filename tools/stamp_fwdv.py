"""Cycle stamps of mlp3_fwd_kernel<9, value> (library built with -DEM_STAMP_FWDV): one wavefront's tile loop in em_value_all."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 300
tr = MAPPOTrainer(dict(ENV_CONFIG, episode_length=T), seed=0, device="cuda:0", n_envs=B, verbose=False)
L = _lib.lib()
assert L.em_debug_fwdv_stamps(None, 1) == 0
tr.train_episode(explore=True, record_buffer=True)
torch.cuda.synchronize()
st = (C.c_ulonglong * 8)()
assert L.em_debug_fwdv_stamps(st, 0) == 0
names = ["", "wait for the tile's inputs", "layer 1 (144 MFMAs)", "prefetch issue", "layer 2 (64 MFMAs)", "head, store, loop"]
n = st[0]
tot = sum(st[i] for i in range(1, 6))
print(f"{n} tiles, {tot / n:.0f} cycles per tile")
for i in range(1, 6):
    print(f"   {names[i]:32s} {st[i] / n:8.0f}  {100 * st[i] / tot:5.1f} %")
