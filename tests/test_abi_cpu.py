"""CPU-side checks of the C-ABI library: it loads without a GPU and exports every symbol that
include/edgemarl.h declares (no compute call is made here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "edgemarl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ems?_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_entry_points():
    names = _declared()
    for must in ("em_create", "em_destroy", "em_reset", "em_step", "em_episode_metrics", "em_last_error"):
        assert must in names


def test_library_loads_and_exports_every_declared_symbol():
    from marl_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libedgemarl.so not built: run __graft_entry__.build()")
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(L, n)]
    assert not missing, missing
    assert _lib.lib().em_abi_version() == _lib.ABI_VERSION


def test_struct_sizes_match_header():
    """The ctypes images must have the C layout (double alignment makes these sizes exact)."""
    from marl_amd import _lib
    from marl_amd.config import EmConfig
    assert ctypes.sizeof(EmConfig) == 16 + 64 * 8 + 21 * 8 + 8 == 704
    assert ctypes.sizeof(_lib.EmTape) == 12 * 8 + 8
    assert ctypes.sizeof(_lib.EmStepOut) == 10 * 8


def test_no_cpu_fallback():
    """The product refuses to run without a GPU device instead of silently computing on the CPU."""
    from marl_amd.env import BatchedEdgeSimGym
    with pytest.raises(RuntimeError):
        BatchedEdgeSimGym(n_envs=2, device="cpu")


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "marl_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert not re.search(r'#include\s*[<"][^>"]*oracle', src), f
                assert "libedge_oracle" not in src and "dlopen" not in src, f


def test_early_stop_threshold_matches_reference_example():
    import run_mappo_batch as rb
    # the reference's own worked example: baseline -1200, ratio 0.8 -> -1440 (run_mappo_batch.py:100-103)
    assert rb._early_stop_threshold(-1200) == -1440.0
    assert rb.EVAL_INTERVAL == 50_000 and rb.EVAL_EPISODES == 5 and rb.EARLY_STOP_WINDOW == 3


def test_argument_errors_come_back_without_a_gpu():
    """Every entry point validates before its first HIP call: bad arguments return a negative code and a message
    (em_last_error) on a machine without a GPU as well."""
    import ctypes as C
    from marl_amd import _lib
    L = _lib.lib()
    assert L.em_step(None, None, None, None, None, None) < 0
    assert b"null handle" in L.em_last_error()
    fake = 0x10000                     # never dereferenced: validation stops the call
    ro = _lib.EmsRollout(8, 64, 4, 6, 5, 0, *([fake] * 20))
    assert L.ems_rollout_index(C.byref(ro), 0, 6, None, None, None, None) < 0 and b"null argument" in L.em_last_error()
    assert L.ems_rollout_index(C.byref(ro), 5, 6, fake, fake, fake, None) < 0 and b"outside" in L.em_last_error()
    assert L.ems_rollout_index(C.byref(ro), 0, 3, fake, fake, fake, None) < 0 and b"n_groups" in L.em_last_error()
    assert L.ems_rollout_index(C.byref(ro), 0, 6, fake, fake + 4, None, None) < 0 and b"16-byte aligned" in L.em_last_error()
    net = _lib.EmGmlp(fake, fake, fake, fake, fake, fake, 0, 68, 64, 65, 2)
    args = [C.byref(net), fake, fake, None, 68, 16, fake, None, 1, None, 1, 0, 0, 0, 0, fake, fake, fake, 64, fake, fake, fake, fake, None]
    assert L.em_gmlp_policy_gather(*args) < 0 and b"null argument" in L.em_last_error()      # x_copy missing
    args[3] = fake
    args[18] = 10
    assert L.em_gmlp_policy_gather(*args) < 0 and b"n_place + 1" in L.em_last_error()
