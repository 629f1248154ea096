"""configs[0] plumbing: the per-configuration loop of run_mappo_batch.py (run_one_cfg) on the HIP trainer."""
import json

import pytest

pytestmark = pytest.mark.gpu


def test_run_one_cfg_log_layout_and_overrides(tmp_path, monkeypatch):
    import run_mappo_batch as rb
    from marl_amd import batch
    from marl_amd.config import ENV_CONFIG
    monkeypatch.setattr(batch, "EVAL_INTERVAL", 80)
    monkeypatch.setattr(batch, "EVAL_EPISODES", 2)
    cfg = {"id": 4, "lr_act": 1e-4, "lr_cri": 8e-4, "batch": 64, "clip": 0.15, "entropy": 0.005, "num_sgd": 6}
    env_cfg = dict(ENV_CONFIG, episode_length=40)
    path = rb.run_one_cfg(cfg, seed=1, max_steps=160, early_stop=True, phase="search", tag="t", n_envs=3,
                          env_config=env_cfg, log_dir=tmp_path)
    data = json.loads(open(path).read())
    assert path.name == "cfg_4_seed_1_search_t.json"
    assert data["total_steps"] == 160 and data["early_stop"] is False and data["cfg"] == cfg
    assert [r["timestep"] for r in data["log"]] == [80, 160]
    for r in data["log"]:
        for k in ("algorithm", "cfg_id", "seed", "phase", "episode_reward", "avg_latency_ms", "p99_latency_ms",
                  "avg_energy_J", "throughput_tps", "load_balance_jain"):
            assert k in r
    # a finished run is skipped, as the reference does
    assert rb.run_one_cfg(cfg, seed=1, max_steps=160, early_stop=True, phase="search", tag="t", n_envs=3,
                          env_config=env_cfg, log_dir=tmp_path) is None

