"""Drop-in module name of the reference's ``run_mappo_batch.py`` (the per-configuration loop ``run_one_cfg`` and the
constants it reads); see ``marl_amd/batch.py``."""
from marl_amd import batch as _batch
from marl_amd.batch import *  # noqa: F401,F403
from marl_amd.batch import early_stop_threshold as _early_stop_threshold  # noqa: F401
from marl_amd.batch import load_baseline_reward as _load_baseline_reward  # noqa: F401
from marl_amd.batch import main, run_one_cfg  # noqa: F401

if __name__ == "__main__":
    main()
