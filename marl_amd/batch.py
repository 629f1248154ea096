"""Per-configuration train/evaluate loop of the reference's sweep harness (``run_mappo_batch.py:243-335``,
``run_one_cfg``) on the batched MI355X trainer.  Same override points (``run_mappo_batch.py:255-261``), same
evaluation cadence, moving-average early stop and JSON log layout, so the reference's log consumers keep working.
The stratified sampling of the search space and the top-k picking of the reference (``:125-241``) are plain Python
with no compute path and are not rebuilt here."""
from __future__ import annotations

import json
import time
from collections import deque
from pathlib import Path

from .config import ENV_CONFIG
from .trainer import MAPPOTrainer

# run_mappo_batch.py:46-66
SEARCH_MAX_STEPS = 2_000_000
FINAL_MAX_STEPS = 3_000_000
EARLY_STOP_STEPS = 1_000_000
EARLY_STOP_RATIO = 0.8
EARLY_STOP_WINDOW = 3
EVAL_INTERVAL = 50_000
EVAL_EPISODES = 5
BASELINE_REWARD = -1200

DEFAULT_CFG = {"id": 0, "lr_act": 3e-4, "lr_cri": 3e-4, "batch": 128, "clip": 0.2, "entropy": 0.01, "num_sgd": 4}


def load_baseline_reward(path="baselines.json"):
    """run_mappo_batch.py:68-83: ``mappo.reward_baseline`` of baselines.json, or None."""
    p = Path(path)
    if not p.exists():
        return None
    try:
        val = json.loads(p.read_text(encoding="utf-8")).get("mappo", {}).get("reward_baseline")
        return None if val is None else float(val)
    except Exception:
        return None


def early_stop_threshold(baseline_reward: float) -> float:
    """run_mappo_batch.py:93-106: baseline minus (1 - ratio) * |baseline| (rewards are negative)."""
    return float(baseline_reward) - (1.0 - float(EARLY_STOP_RATIO)) * abs(float(baseline_reward))


def run_one_cfg(cfg: dict, seed: int, max_steps: int, early_stop: bool, phase: str, tag: str | None = None,
                n_envs: int = 1, device: str = "cuda:0", env_config: dict | None = None, log_dir="hparam_logs/mappo"):
    """One configuration: train until ``max_steps`` environment steps (per environment), evaluate every
    ``EVAL_INTERVAL`` steps, stop early on the moving mean of the training reward, write the JSON log.
    Returns the path of the log (None when it already existed, as the reference skips finished runs)."""
    log_dir = Path(log_dir)
    log_dir.mkdir(parents=True, exist_ok=True)
    out_path = log_dir / f"cfg_{cfg['id']}_seed_{seed}_{phase}{'' if not tag else '_' + tag}.json"
    if out_path.exists():
        print(f"[SKIP] {out_path} exists")
        return None

    trainer = MAPPOTrainer(ENV_CONFIG if env_config is None else env_config, seed=seed, device=device, n_envs=n_envs)
    m = trainer.mappo
    m.actor_optimizer.param_groups[0]["lr"] = cfg["lr_act"]
    m.critic_optimizer.param_groups[0]["lr"] = cfg["lr_cri"]
    m.clip_param = cfg["clip"]
    m.entropy_coeff = cfg["entropy"]
    m.num_sgd_iter = cfg["num_sgd"]
    m.mini_batch_size = cfg["batch"]        # inert in the reference too (no reader)

    results = []
    total_steps = 0
    t0 = time.time()
    stopped, reason = False, ""
    recent = deque(maxlen=EARLY_STOP_WINDOW)
    threshold = None if BASELINE_REWARD is None else early_stop_threshold(BASELINE_REWARD)

    while total_steps < max_steps:
        ep = trainer.train_episode(explore=True, record_buffer=True)
        total_steps += ep["episode_steps"]
        if total_steps % EVAL_INTERVAL == 0 or total_steps >= max_steps:
            ev = trainer.evaluate(num_episodes=EVAL_EPISODES)
            results.append({"algorithm": "MAPPO", "cfg_id": cfg["id"], "seed": seed, "phase": phase,
                            "timestep": total_steps, "episode_reward": ep["episode_reward"], **ev})
            recent.append(ep["episode_reward"])
            print(f"[{phase}] cfg={cfg['id']} seed={seed} steps={total_steps} reward={ep['episode_reward']:.2f} "
                  f"latency={ev['avg_latency_ms']:.2f}")
            if (early_stop and BASELINE_REWARD is not None and total_steps >= EARLY_STOP_STEPS
                    and len(recent) == EARLY_STOP_WINDOW and sum(recent) / EARLY_STOP_WINDOW < threshold):
                stopped = True
                reason = (f"moving_avg_reward<threshold (ratio={EARLY_STOP_RATIO}, baseline={BASELINE_REWARD}) over "
                          f"last {EARLY_STOP_WINDOW} evals")
                break
        m.update()

    with open(out_path, "w", encoding="utf-8") as f:
        json.dump({"cfg": cfg, "seed": seed, "phase": phase, "max_steps": max_steps, "total_steps": total_steps,
                   "early_stop": stopped, "early_stop_reason": reason, "time_sec": time.time() - t0,
                   "n_envs": n_envs, "log": results}, f, indent=2)
    print(f"[SAVE] {out_path} total steps {total_steps}")
    return out_path


def main(argv=None):
    """``python run_mappo_batch.py [final] [--tag X] [--cfg-json FILE] [--envs B] [--max-steps N]``: runs the
    configurations of FILE (a JSON list of dicts with the reference's keys) or the reference's default
    hyper-parameters, seed = cfg id % 3 as in the reference (:413-416)."""
    import argparse
    import sys
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", nargs="?", default="search")
    ap.add_argument("--tag", default=None)
    ap.add_argument("--cfg-json", default=None)
    ap.add_argument("--envs", type=int, default=1)
    ap.add_argument("--max-steps", type=int, default=None)
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(sys.argv[1:] if argv is None else argv)
    global BASELINE_REWARD
    loaded = load_baseline_reward()
    if loaded is not None:
        BASELINE_REWARD = loaded
    final = a.mode.lower() == "final"
    space = json.loads(Path(a.cfg_json).read_text()) if a.cfg_json else [dict(DEFAULT_CFG)]
    for cfg in space:
        run_one_cfg(cfg, cfg["id"] % 3, max_steps=a.max_steps or (FINAL_MAX_STEPS if final else SEARCH_MAX_STEPS),
                    early_stop=not final, phase="final" if final else "search", tag=a.tag, n_envs=a.envs, device=a.device)
