"""Drop-in module name of the reference's environment file (``MRRL.PY``; importable on Linux as ``MRRL``).
Everything is re-exported from ``marl_amd``: the dict-API ``EdgeSimGym`` is the n_envs=1 view of the HIP
kernels, ``BatchedEdgeSimGym`` is the same environment with a leading batch dimension."""
from marl_amd.baselines import run_greedy_baseline  # noqa: F401
from marl_amd.config import ENV_CONFIG  # noqa: F401
from marl_amd.dict_env import EdgeSimGym  # noqa: F401
from marl_amd.env import BatchedEdgeSimGym  # noqa: F401
from marl_amd.experiment import run_experiment  # noqa: F401
from marl_amd.trainer import run_mappo_experiment  # noqa: F401

MAPPO_AVAILABLE = True
