"""em_gae_static (the entry point that allocates nothing: its scratch is a fixed array inside the library) against
em_gae (caller-owned scratch) through the C ABI: advantages, returns and the three moments must be IDENTICAL bit for bit
(same kernels, same summation order), on two calls in a row, the second on a smaller batch so that the scratch still
holds the first call's partial rows.  Collected last on purpose (file name): a failure here hides no other test."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _storage(B, T, seed, device):
    """A recorded rollout built directly in the em_rollout buffers: random acting-agent counts per (step, env), entry
    numbering in (step, env, rank) order as em_rollout_index produces it, random rewards/values, done on the last row."""
    from marl_amd.mappo import RolloutStorage
    rs = np.random.RandomState(seed)
    ro = RolloutStorage(B, 10, T, device)
    k = rs.choice([0, 0, 1, 1, 2, 3, 5], size=(T, B)).astype(np.uint8)
    first = (np.cumsum(k.reshape(-1).astype(np.int64)) - k.reshape(-1)).reshape(T, B).astype(np.int32)
    done = np.zeros((T, B), np.uint8)
    done[T - 1] = 1
    ro.t["k_tb"].copy_(torch.as_tensor(k))
    ro.t["first_tb"].copy_(torch.as_tensor(first))
    ro.t["done_tb"].copy_(torch.as_tensor(done))
    ro.t["reward_tb"].copy_(torch.as_tensor(-rs.gamma(2.0, 3.0, size=(T, B))))
    ro.t["value_tb"].copy_(torch.as_tensor(rs.normal(-20.0, 8.0, size=(T, B)).astype(np.float32)))
    return ro, int(k.sum())


def _run(L, ro, E, static, device):
    from marl_amd import _lib
    ro.t["ent_adv"].fill_(float("nan"))
    ro.t["ent_ret"].fill_(float("nan"))
    mom = torch.full((3,), float("nan"), dtype=torch.float64, device=device)
    s = int(torch.cuda.current_stream(device).cuda_stream)
    if static:
        _lib.check(L.em_gae_static(ro.ref, 0.99, 0.95, mom.data_ptr(), s), "em_gae_static")
    else:
        part = torch.zeros(3 * ((ro.n_envs + 255) // 256), dtype=torch.float64, device=device)
        _lib.check(L.em_gae(ro.ref, 0.99, 0.95, part.data_ptr(), part.numel(), mom.data_ptr(), s), "em_gae")
    torch.cuda.synchronize(device)
    return ro.t["ent_adv"][:E].cpu().numpy(), ro.t["ent_ret"][:E].cpu().numpy(), mom.cpu().numpy()


def test_gae_static_equals_gae_on_two_calls_second_smaller():
    from marl_amd import _lib
    L = _lib.lib()
    device = torch.device("cuda:0")
    for B, T, seed in ((3000, 37, 1), (517, 23, 2)):        # 12 workgroups of partials, then 3: stale rows 3..11 remain
        ro, E = _storage(B, T, seed, device)
        want = _run(L, ro, E, False, device)
        got = _run(L, ro, E, True, device)
        assert want[2][0] == E and np.isfinite(want[0]).all()
        for w, g, name in zip(want, got, ("advantages", "returns", "moments")):
            assert np.array_equal(w, g), f"{name} differ at n_envs={B}"


def test_gae_static_rejects_sizes_beyond_its_scratch():
    from marl_amd import _lib
    L = _lib.lib()
    ro = _lib.EmRollout()
    ro.n_envs, ro.horizon = (1 << 20) + 1, 1
    mom = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    rc = L.em_gae_static(C.byref(ro), 0.99, 0.95, mom.data_ptr(), None)
    assert rc == -1 and b"fixed scratch" in L.em_last_error()
