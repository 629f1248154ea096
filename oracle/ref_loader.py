"""Container-only tool: load the reference implementation for golden-vector generation.

TEST INFRASTRUCTURE. Nothing under tests/, bench.py or the product imports this module; only
``oracle/make_golden.py`` does, and only inside the build container where ``/root/reference`` is
mounted.  The reference never travels to the GPU box.

``MRRL.PY`` has an upper-case suffix (invisible to Linux ``import``) and top-level imports of
packages that are not installed here (gymnasium, ray, seaborn).  None of them takes part in the
arithmetic of ``EdgeSimGym`` (they provide a base class and space descriptors), so they are
replaced by empty stand-in modules before the file's source text is compiled and executed (never its shipped .pyc).
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _stub(name: str, **attrs) -> types.ModuleType:
    mod = types.ModuleType(name)
    mod.__dict__.update(attrs)
    sys.modules[name] = mod
    return mod


class _AnySpace:
    def __init__(self, *args, **kwargs):
        pass


class _BaseEnv:
    def __init__(self):
        pass


def load_reference():
    """Returns (MRRL module, mappo_algorithm module, mappo_trainer module)."""
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError("reference tree not present; goldens can only be regenerated in the build container")
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    if "MRRL" not in sys.modules or not hasattr(sys.modules["MRRL"], "_is_reference"):
        spaces = _stub("gymnasium.spaces", Box=_AnySpace, Discrete=_AnySpace, Dict=_AnySpace)
        _stub("gymnasium", spaces=spaces)
        _stub("ray")
        _stub("ray.rllib")
        _stub("ray.rllib.env")
        _stub("ray.rllib.env.multi_agent_env", MultiAgentEnv=_BaseEnv)
        _stub("ray.rllib.algorithms")
        _stub("ray.rllib.algorithms.ppo", PPOConfig=object)
        _stub("ray.rllib.algorithms.callbacks", DefaultCallbacks=object)
        dep = _stub("ray.rllib.utils.deprecation", deprecation_warning=lambda *a, **k: None)
        _stub("ray.rllib.utils", deprecation=dep)
        _stub("ray.tune")
        _stub("ray.tune.registry", register_env=lambda *a, **k: None)
        _stub("seaborn")
        _exec_source("MRRL", "MRRL.PY")._is_reference = True
    # the reference's own modules are executed from their SOURCE TEXT only (compile + exec): the tree also ships
    # __pycache__/*.cpython-310.pyc files, and a path-based import would run that prebuilt bytecode whenever its header
    # happened to match the source's mtime/size.  The reference root is never put on sys.path.
    for name in ("mappo_algorithm", "mappo_trainer"):
        if not getattr(sys.modules.get(name), "_is_reference", False):
            _exec_source(name, name + ".py")._is_reference = True
    return sys.modules["MRRL"], sys.modules["mappo_algorithm"], sys.modules["mappo_trainer"]


def _exec_source(name: str, filename: str) -> types.ModuleType:
    path = os.path.join(REFERENCE_ROOT, filename)
    with open(path, "r", encoding="utf-8") as f:
        code = compile(f.read(), path, "exec")
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod          # `from MRRL import ...` / `from mappo_algorithm import MAPPO` resolve through here
    exec(code, mod.__dict__)
    return mod


class RecordingRandom:
    """Stands in for ``numpy.random`` inside the reference module and logs every legacy draw.

    Each log entry is ``(name, args, result)``.  Forwarding goes to the real global RandomState so
    seeds behave exactly as in the reference.
    """

    _NAMES = ("uniform", "poisson", "exponential", "randint", "rand")

    def __init__(self, np_module):
        self._np = np_module
        self.log = []

    def seed(self, s):
        self._np.random.seed(s)

    def __getattr__(self, name):
        fn = getattr(self._np.random, name)
        if name not in self._NAMES:
            return fn

        def wrapped(*args, **kwargs):
            out = fn(*args, **kwargs)
            self.log.append((name, tuple(float(a) for a in args), float(out)))
            return out

        return wrapped


class NumpyProxy:
    """``MRRL.np`` replacement: numpy with ``.random`` swapped for a RecordingRandom."""

    def __init__(self, np_module):
        self._np = np_module
        self.random = RecordingRandom(np_module)

    def __getattr__(self, name):
        return getattr(self._np, name)
