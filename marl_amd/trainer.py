"""MAPPOTrainer on MI355X: the reference trainer's interface (``mappo_trainer.py:23-394``: constructor,
``train_episode``, ``train``, ``evaluate``, ``save_model/load_model``, ``run_mappo_experiment``) with a leading
batch-of-environments dimension (``n_envs``).  One call of ``train_episode`` plays one episode in EVERY
environment; ``trainer.mappo.update()`` then consumes the whole batch exactly as the reference consumes one
episode buffer.  Multi-GPU: one process per GPU, environments sharded by rank, gradients all-reduced in
``MAPPO.update`` (RCCL)."""
from __future__ import annotations

import sys

import numpy as np
import torch

from .config import ENV_CONFIG, mean_arrivals_bound, summarise_episode_metrics
from .env import METRIC_KEYS, BatchedEdgeSimGym
from .mappo import MAPPO, RolloutStorage
from .parallel import GradSync

_STAT_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "load_balance_jain")


class MAPPOTrainer:
    def __init__(self, env_config, seed=0, device="cuda:0", n_envs=1, rank=0, world_size=1, process_group=None,
                 verbose=True, update_interval=1, capacity_entries=None):
        if str(device) == "cuda":
            device = "cuda:0"
        self.device = torch.device(device)
        self.seed, self.n_envs, self.rank, self.world_size = int(seed), int(n_envs), int(rank), int(world_size)
        cfg = {**env_config}
        self.env_offset = self.rank * self.n_envs
        self.env = BatchedEdgeSimGym(cfg, n_envs=self.n_envs, device=self.device, seed=self.seed,
                                     env_offset=self.env_offset)
        # same seeding calls as the reference (mappo_trainer.py:43-46); torch's CPU generator then gives the
        # reference's initial weights for this seed
        np.random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed(seed)
        self.n_agents = int(cfg["n_nodes"])
        # the reference probes up to ten reseeded resets for a non-empty observation (mappo_trainer.py:53-58),
        # which leaves the environment stream keyed by seed + attempt
        # Data parallel: the probe stops on the OR over ALL ranks of "some environment has an observation" - the condition
        # the one-process run on the union of the shards evaluates - so every rank leaves the loop at the same
        # seed + attempt (a rank-local test could stop two small shards at different attempts, i.e. on different streams).
        probe = GradSync(process_group)
        for attempt in range(10):
            _, active = self.env.reset(seed=self.seed + attempt)
            if probe.sum_counts([int(bool((active != 0).any()))], self.device)[0] > 0:
                break
        self.reset_attempts = attempt + 1
        self.obs_dim = 5 + (self.n_agents - 1)
        self.action_dim = self.n_agents + 1
        self.state_dim = self.obs_dim * self.n_agents
        self.mappo = MAPPO(obs_dim=self.obs_dim, state_dim=self.state_dim, action_dim=self.action_dim,
                           n_agents=self.n_agents, lr=3e-4, gamma=0.99, gae_lambda=0.95, clip_param=0.2,
                           value_loss_coeff=0.5, entropy_coeff=0.01, max_grad_norm=0.5, num_sgd_iter=4,
                           device=self.device, n_envs=self.n_envs,
                           horizon=self.env.episode_len * max(1, int(update_interval)), process_group=process_group,
                           capacity_entries=capacity_entries, entries_per_step=mean_arrivals_bound(cfg))
        self.update_interval = max(1, int(update_interval))   # episodes the rollout storage can hold before update()
        B, n = self.n_envs, self.n_agents
        self.placement = torch.zeros((B, n), dtype=torch.int32, device=self.device)
        self.resource = torch.full((B, n), 0.5, dtype=torch.float32, device=self.device)
        self._train_storage = self.mappo.buffer
        self._scratch_storage = None
        self.policy_episode = 0
        self.overlap_value = True      # value forward on a side stream during recorded rollouts (identical results)
        # measurement hook (bench.py): a list here makes train_episode record three events per step on the launch stream -
        # before the index/actor launch, between it and the environment step, after the step
        self.profile_events = None
        if verbose and self.rank == 0:
            print(f"[OK] MAPPO (MI355X) initialised: obs {self.obs_dim}, state {self.state_dim}, action {self.action_dim}, "
                  f"agents {self.n_agents}, envs/GPU {self.n_envs}, device {self.device}")

    # ------------------------------------------------------------------------------------------------------
    def train_episode(self, explore=True, record_buffer=True, tapes=None, noise=None):
        """One episode in every environment (mappo_trainer.py:177-250).  ``tapes`` (list of DeviceTape, one per
        reset/step) and ``noise`` ([entries, action_dim] standard normal draws) replace the in-kernel random
        streams for parity runs.  Returns the reference's stat keys (means over the environments) plus per-env
        arrays under ``per_env``."""
        env, m = self.env, self.mappo
        T = env.episode_len
        if record_buffer:
            m.buffer = self._train_storage
        else:   # evaluation episodes must not clobber a recorded batch (run_mappo_batch.py:272-317 evaluates first)
            if self._scratch_storage is None:
                self._scratch_storage = RolloutStorage(self.n_envs, self.n_agents, 1, self.device, by_reference=True)
            m.buffer = self._scratch_storage
        ro = m.buffer
        # like the reference's list buffer, recorded episodes accumulate until update() clears them
        # (mappo_trainer.py:281-282); the storage holds update_interval episodes
        base = ro.steps_recorded if record_buffer else 0
        if record_buffer and base + T > ro.horizon:
            raise RuntimeError(f"rollout storage holds {ro.horizon // T} episode(s): call mappo.update() or construct "
                               "MAPPOTrainer with a larger update_interval")
        # The environment writes the observations of step t + 1 straight into slice t + 1 of the storage's observation
        # history (states are held by reference, nothing is copied) and counts the acting agents per 256-environment chunk
        # on the way out; the two total buffers alternate, em_rollout_index_totals clears the one the next step adds into.
        tot = ro.chunk_totals
        tot.zero_()
        par = 0
        own_obs = env.obs
        env.bind_obs(ro.obs_slice(base) if ro.by_reference else None, tot[par])
        env.reset(tape=tapes[0] if tapes is not None else None)
        if base == 0:
            m.begin_rollout()
        self.policy_episode += 1
        total_reward = torch.zeros(self.n_envs, dtype=torch.float64, device=self.device)
        m.overlap_value = bool(record_buffer) and self.overlap_value
        try:
            prof = self.profile_events
            stream = torch.cuda.current_stream(self.device) if prof is not None else None
            for t in range(T):
                slot = base + t if record_buffer else 0
                if prof is not None:
                    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                    ev[0].record(stream)
                m.record_and_act(slot, env.obs, env.active, env.rank, self.placement, self.resource, explore=explore,
                                 noise=noise, seed=self.seed, episode=self.policy_episode, env_offset=self.env_offset,
                                 rng_step=t, totals=tot[par], totals_next=tot[par ^ 1])
                par ^= 1
                if prof is not None:
                    ev[1].record(stream)
                env.bind_obs(ro.obs_slice(base + t + 1) if ro.by_reference else None, tot[par])
                if record_buffer:
                    env.step(self.placement, self.resource, tape=tapes[t + 1] if tapes is not None else None,
                             reward_seen_out=ro.t["reward_tb"][slot], done_out=ro.t["done_tb"][slot])
                else:
                    env.step(self.placement, self.resource, tape=tapes[t + 1] if tapes is not None else None)
                    total_reward += env.reward
                if prof is not None:
                    ev[2].record(stream)
                    prof.append(ev)
        finally:
            m.overlap_value = False
            m.join_value()
            m.buffer = self._train_storage
            # the environment keeps its last observation in the storage's slice: callers that step it by hand afterwards get
            # its own buffer back (with the current observation in it)
            if ro.by_reference:
                own_obs.copy_(env.obs)
            env.bind_obs(own_obs if ro.by_reference else None, None)
        if record_buffer:
            m.compute_values()            # deferred critic: one launch over the recorded states (a no-op in the per-step modes)
            total_reward = ro.t["reward_tb"][base:base + T].sum(dim=0)
        metrics = env.episode_metrics()                      # [B, 6] on the device
        # the reference harvests the metrics only when the final step produced observations
        # (mappo_trainer.py:222-231), otherwise it reports 0.0 for all of them
        mean, p99_missing = summarise_episode_metrics(metrics, env.active != 0)
        idx = {k: i for i, k in enumerate(METRIC_KEYS)}
        stats = {"episode_reward": float(total_reward.mean().item()), "episode_steps": T}
        for k in _STAT_KEYS:
            stats[k] = float(mean[idx[k]])
        stats["p99_envs_missing"] = p99_missing      # environments whose latency log overflowed (their p99 is NaN, not 0)
        stats["per_env"] = {"episode_reward": total_reward, "metrics": metrics}
        if record_buffer:
            # a full storage is reported here, where the episode's statistics are fetched anyway; the error itself is raised by
            # mappo.update() on ALL ranks together (a rank-local raise in front of a collective would strand the others)
            e, s_, err = ro.totals_raw()
            stats["storage_overflow"] = bool(err)
            if err:
                print("[marl_amd] " + ro.overflow_message(e, s_) + "; this batch is incomplete and update() will raise",
                      file=sys.stderr, flush=True)
        return stats

    # -- the reference's dict-level helpers (mappo_trainer.py:101-175), one environment at a time -----------------------------
    def obs_dict_to_state(self, obs_dict):
        """mappo_trainer.py:101-123: agent-major concatenation, zero rows for agents without an observation.  (The batched
        path never builds this on the host: the dense observation tensor of the step kernel IS this state.)"""
        state = np.zeros((self.n_agents, self.obs_dim), dtype=np.float32)
        for i in range(self.n_agents):
            o = obs_dict.get(f"node_{i}")
            if o is not None:
                state[i] = np.asarray(o, np.float32).reshape(-1)
        return state.reshape(-1)

    def _format_env_actions(self, actions_dict):
        """mappo_trainer.py:125-158: raw action vectors -> ``{"placement": int, "resource": float32[1]}`` per agent
        (``em_format_actions``: first argmax unless all logits are ~0, tanh-squashed share in float64)."""
        from . import _lib
        ids = list(actions_dict.keys())
        if not ids:
            return {}
        A = self.n_agents + 1
        raw = np.zeros((len(ids), A), np.float32)
        for i, a in enumerate(ids):
            vec = np.asarray(actions_dict[a], np.float32).reshape(-1)
            if vec.size < A:                       # short vectors are zero-padded; the share is then read from the pad
                raw[i, :vec.size] = vec
            else:                                  # the reference reads logits [0, n) and the LAST component
                raw[i, :self.n_agents], raw[i, self.n_agents] = vec[:self.n_agents], vec[-1]
        x = torch.as_tensor(raw).to(self.device)
        pl = torch.empty(len(ids), dtype=torch.int32, device=self.device)
        rs = torch.empty(len(ids), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().em_format_actions(x.data_ptr(), len(ids), self.n_agents, pl.data_ptr(), rs.data_ptr(),
                                                    int(torch.cuda.current_stream(self.device).cuda_stream)), "em_format_actions")
        pl, rs = pl.cpu().numpy(), rs.cpu().numpy()
        return {a: {"placement": int(pl[i]), "resource": np.array([rs[i]], dtype=np.float32)} for i, a in enumerate(ids)}

    def _store_transition(self, obs_dict, actions_dict, log_probs_dict, state, reward, dones, value):
        """mappo_trainer.py:160-175: one buffer entry per acting agent, all sharing state/reward/done/value."""
        done_flag = dones.get("__all__", False)
        for agent_id, obs in obs_dict.items():
            if agent_id not in actions_dict or agent_id not in log_probs_dict:
                continue
            self.mappo.buffer.add(obs=obs, state=state, action=actions_dict[agent_id], reward=reward, done=done_flag,
                                  log_prob=log_probs_dict[agent_id], value=value)

    def train(self, num_iterations=1000, update_interval=1, eval_interval=10, eval_episodes=10, max_total_steps=5_000_000):
        """mappo_trainer.py:252-316.  ``timestep`` counts environment steps of ONE environment (the reference's
        unit); the batch plays n_envs x world_size times as many."""
        if update_interval > self.update_interval:
            raise ValueError(f"construct MAPPOTrainer(update_interval={update_interval}) so the device rollout storage "
                             "can hold that many episodes")
        results_log, total_steps, iteration = [], 0, 0
        while True:
            iteration += 1
            episode_stats = self.train_episode(explore=True, record_buffer=True)
            total_steps += episode_stats["episode_steps"]
            if iteration % update_interval == 0:            # mappo_trainer.py:281-282
                self.mappo.update()
            if iteration % eval_interval == 0:
                eval_stats = self.evaluate(num_episodes=eval_episodes)
                entry = {"algorithm": "MAPPO", "seed": self.seed, "iteration": iteration, "timestep": total_steps,
                         "episode_return_mean": episode_stats["episode_reward"]}
                entry.update({k: eval_stats[k] for k in _STAT_KEYS})
                results_log.append(entry)
                if self.rank == 0:
                    print(f"Iter {iteration}/{num_iterations} | Steps: {total_steps} | "
                          f"Reward: {episode_stats['episode_reward']:.2f} | Latency: {eval_stats['avg_latency_ms']:.2f}ms")
            if total_steps >= max_total_steps:
                break
        return results_log

    def evaluate(self, num_episodes=100):
        """mappo_trainer.py:318-351: deterministic policy, environment exploration disabled.  Each call of
        train_episode already plays n_envs episodes, so ceil(num_episodes / n_envs) batches are run."""
        prev = self.env.disable_env_exploration
        self.env.disable_env_exploration = True
        acc = {k: [] for k in _STAT_KEYS}
        try:
            for _ in range(max(1, -(-int(num_episodes) // self.n_envs))):
                stats = self.train_episode(explore=False, record_buffer=False)
                for k in _STAT_KEYS:
                    acc[k].append(stats[k])
        finally:
            self.env.disable_env_exploration = prev
        return {k: float(np.mean(v)) for k, v in acc.items()}

    def save_model(self, path):
        self.mappo.save(path)

    def load_model(self, path):
        self.mappo.load(path)


def run_mappo_experiment(seed, num_iterations=1000, max_total_steps=5_000_000, n_envs=1, device="cuda:0"):
    """mappo_trainer.py:364-394."""
    trainer = MAPPOTrainer(ENV_CONFIG, seed=seed, device=device, n_envs=n_envs)
    log = trainer.train(num_iterations=num_iterations, update_interval=1, eval_interval=10, eval_episodes=10,
                        max_total_steps=max_total_steps)
    trainer.save_model(f"mappo_seed_{seed}.pt")
    return log
