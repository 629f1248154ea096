"""Drop-in module name of the reference's ``mappo_trainer.py``; see ``marl_amd/trainer.py``."""
from marl_amd.trainer import MAPPOTrainer, run_mappo_experiment  # noqa: F401
