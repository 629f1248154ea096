"""Cycle stamps of mlp3_bwd_kernel (library built with -DEM_STAMP_BWD): cycles per phase of one activation wavefront, per tile."""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.config import ENV_CONFIG  # noqa: E402
from marl_amd.trainer import MAPPOTrainer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tr = MAPPOTrainer(dict(ENV_CONFIG, episode_length=T), seed=0, device="cuda:0", n_envs=B, verbose=False)
L = _lib.lib()
tr.train_episode(explore=True, record_buffer=True)
torch.cuda.synchronize()
assert L.em_debug_bwd_stamps(None, 1) == 0
tr.mappo.update()
torch.cuda.synchronize()
st = (C.c_ulonglong * 32)()
assert L.em_debug_bwd_stamps(st, 0) == 0
names = ["", "top (flags, stage issue, loss inputs)", "forward", "loss", "dz2, dz1 (+ transposes of dY, H2)", "dW3 (+ dZ2 out)",
         "tr(dZ2) back, H1 out/back", "dW2 (+ dZ1 out)", "dW1 / hand-over + end wait"]
for which, label in ((0, "critic"), (1, "actor")):
    n = st[which * 16]
    if not n:
        continue
    tot = sum(st[which * 16 + i] for i in range(1, 10))
    print(f"{label}: {n} tiles, {tot / n:.0f} cycles per tile")
    for i in range(1, 9):
        v = st[which * 16 + i] / n
        print(f"   {names[i]:45s} {v:8.0f}  {100 * v * n / tot:5.1f} %")
    if st[which * 16 + 9]:
        print(f"   (of the top: up to the end of the wait for the staged tile {st[which * 16 + 9] / n:.0f})")
    print("   SIMD id of the activation / gradient wavefront of pair 0 (+100):", st[which * 16 + 14], st[which * 16 + 15])
    print("   spins per tile waiting for {tiles staged, -, dZ1 published, dZ1 read}:", [round(st[which * 16 + 10 + i] / n, 2) for i in range(4)])
