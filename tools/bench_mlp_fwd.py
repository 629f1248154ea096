"""Forward MLP kernel on dense rows (em_mlp3_forward), critic shape: how much of em_value_all's time is the products."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from marl_amd import _lib  # noqa: E402
from marl_amd.mappo import MAPPOCritic, _stream  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 33_000_000
in_dim = int(sys.argv[2]) if len(sys.argv) > 2 else 140
dev = torch.device("cuda:0")
torch.manual_seed(0)
critic = MAPPOCritic(in_dim).to(dev)
x = torch.randn(rows, in_dim, device=dev)
y = torch.empty(rows, 1, device=dev)
net = critic.net()
L = _lib.lib()
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.check(L.em_mlp3_forward(C.byref(net), x.data_ptr(), in_dim, rows, y.data_ptr(), _stream(dev)), "fwd")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"rows={rows} in_dim={in_dim}: {dt * 1e3:.2f} ms, {rows * 26240 / dt / 1e12:.1f} TFLOP/s, {rows * in_dim * 4 / dt / 1e12:.2f} TB/s")
