"""Cycle stamps of rollout_policy_kernel (library built with -DEM_STAMP_FWD): phases of workgroup 1, wavefront 0."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
tr = MAPPOTrainer(dict(ENV_CONFIG, episode_length=60), seed=0, device="cuda:0", n_envs=B, verbose=False)
L = _lib.lib()
names = ["start -> scan barrier (loads, weights to LDS)", "counters, s_first/s_rank -> barrier", "entry list -> barrier", "entry copy + step records",
         "weight fragments", "tile 1 MFMAs", "tile 1 epilogue", "rest (tile 2 ...)"]
acc = [0.0] * 8
R = 5
for _ in range(R):
    tr.mappo.buffer.clear()
    tr.train_episode(explore=True, record_buffer=True)
    torch.cuda.synchronize()
    st = (C.c_ulonglong * 16)()
    assert L.em_debug_fwd_stamps(st) == 0
    for i in range(8):
        acc[i] += (st[i + 1] - st[i]) / R
print('  epilogue (last tile): box-muller %d, sample+logp %d, shuffles %d, share %d' % tuple((st[i+1]-st[i]) for i in range(9,13)))
tot = 0
for n_, v in zip(names, acc):
    print(f"  {n_:50s} {v:9.0f} cycles")
    tot += v
print(f"  total {tot:.0f} cycles")
