"""MFMA fused-MLP forward vs a plain PyTorch fp32 reference (exact on small-integer data, 1e-5 on random)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(in_dim, out_dim, integer, seed):
    g = torch.Generator().manual_seed(seed)
    if integer:
        mk = lambda *s: torch.randint(-2, 3, s, generator=g).float()   # noqa: E731
    else:
        mk = lambda *s: (torch.rand(*s, generator=g) - 0.5) * 0.5       # noqa: E731
    return {"w1": mk(64, in_dim), "b1": mk(64), "w2": mk(64, 64), "b2": mk(64), "w3": mk(out_dim, 64), "b3": mk(out_dim)}


def _ref(p, x):
    h = torch.relu(x @ p["w1"].T + p["b1"])
    h = torch.relu(h @ p["w2"].T + p["b2"])
    return h @ p["w3"].T + p["b3"]


@pytest.mark.parametrize("in_dim,out_dim", [(14, 11), (140, 1), (68, 16), (16, 3), (1, 1)])
@pytest.mark.parametrize("rows", [1, 16, 37, 4099])
@pytest.mark.parametrize("integer", [True, False])
def test_mlp3_forward(in_dim, out_dim, rows, integer):
    from marl_amd import _lib
    L = _lib.lib()
    p = _net(in_dim, out_dim, integer, seed=in_dim * 100 + out_dim)
    g = torch.Generator().manual_seed(rows)
    x = torch.randint(-3, 4, (rows, in_dim), generator=g).float() if integer else torch.randn(rows, in_dim, generator=g)
    want = _ref(p, x)
    d = {k: v.cuda().contiguous() for k, v in p.items()}
    net = _lib.EmMlp3(d["w1"].data_ptr(), d["b1"].data_ptr(), d["w2"].data_ptr(), d["b2"].data_ptr(),
                      d["w3"].data_ptr(), d["b3"].data_ptr(), in_dim, 64, out_dim)
    xd = x.cuda().contiguous()
    y = torch.full((rows, out_dim), float("nan"), device="cuda")
    _lib.check(L.em_mlp3_forward(C.byref(net), xd.data_ptr(), in_dim, rows, y.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream))
    got = y.cpu()
    if integer:
        assert torch.equal(got, want)          # small integers: every product and sum is exact in f32
    else:
        torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-5)
