"""Cycle stamps of gmlp_fwd_kernel's policy variant (library built with -DEM_STAMP_GFWD): phases of every wavefront of one workgroup."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.config import SCALED_ENV_CONFIG  # noqa: E402
from marl_amd.ippo import IPPOTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
alg = sys.argv[2] if len(sys.argv) > 2 else "ippo"
tr = IPPOTrainer(dict(SCALED_ENV_CONFIG, episode_length=40), seed=0, device="cuda:0", n_envs=B, algorithm=alg, verbose=False)
L = _lib.lib()
for _ in range(3):
    tr.algo.buffer.clear()
    tr.train_episode(explore=True, record_buffer=True)
    torch.cuda.synchronize()
st = (C.c_ulonglong * 128)()
assert L.em_debug_gfwd_stamps(st) == 0
t0 = min(st[w * 8] for w in range(16))
print("wave: start | staged | tile start | x loaded | products done | epilogue done | end   (cycles from the first wavefront's start)")
for w in range(16):
    print(w, [int(st[w * 8 + i]) - int(t0) if st[w * 8 + i] else None for i in range(7)])
