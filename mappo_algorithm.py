"""Drop-in module name of the reference's ``mappo_algorithm.py``; see ``marl_amd/mappo.py``."""
from marl_amd.mappo import MAPPO, MAPPOActor, MAPPOCritic, RolloutBuffer, RolloutStorage, ValueNormalizer  # noqa: F401
