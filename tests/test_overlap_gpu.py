"""GPU: the value forward on a side stream (trainers' overlap_value) changes nothing: buffers after a recorded episode and
parameters after the update are bit-identical to the serial order, for MAPPO, IPPO and MAPPO on the scaled topology."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _main(overlap):
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG)
    cfg["episode_length"] = 120
    tr = MAPPOTrainer(cfg, seed=4, device=DEV, n_envs=512, verbose=False)
    tr.overlap_value = overlap
    out = []
    for _ in range(2):
        tr.train_episode(explore=True, record_buffer=True)
        ro = tr.mappo.buffer
        out.append(ro.t["value_tb"][:120].clone())
        out.append(ro.t["st_value"].clone())
        tr.mappo.update()
        out.append(tr.mappo.flat_params.detach().clone())
    return out


def _scaled(algorithm, overlap):
    from marl_amd.config import SCALED_ENV_CONFIG
    from marl_amd.ippo import IPPOTrainer
    cfg = dict(SCALED_ENV_CONFIG)
    cfg.update(n_nodes=8, n_deciders=24, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9, episode_length=60)
    tr = IPPOTrainer(cfg, seed=6, device=DEV, n_envs=96, verbose=False, algorithm=algorithm)
    tr.overlap_value = overlap
    out = []
    for _ in range(2):
        tr.train_episode(explore=True, record_buffer=True)
        tt = tr.algo.buffer.t
        out.append(tt["value_tb"].clone() if algorithm == "mappo" else tt["ent_value"].clone())
        tr.algo.update()
        out.append(tr.algo.actor.flat.clone())
        out.append(tr.algo.critic.flat.clone())
    return out


def test_main_mappo_overlap_is_bit_identical():
    a, b = _main(True), _main(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("algorithm", ["ippo", "mappo"])
def test_scaled_overlap_is_bit_identical(algorithm):
    a, b = _scaled(algorithm, True), _scaled(algorithm, False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
