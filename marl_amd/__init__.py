"""marl_amd - MI355X-native batched EdgeSimGym + MAPPO hot path.

The compute lives in hand-written gfx950 HIP kernels (marl_amd/csrc, libedgemarl.so) reached
through the C ABI of include/edgemarl.h; this package is the thin Python host side that mirrors the
reference's API (MRRL.EdgeSimGym, mappo_algorithm.MAPPO, mappo_trainer.MAPPOTrainer) with a leading
batch-of-environments dimension.  There is no CPU fallback: importing the kernels without the built
library raises.
"""
from .config import ENV_CONFIG, make_em_config  # noqa: F401

__all__ = ["ENV_CONFIG", "make_em_config"]
