"""The reference's one-environment dict/list API (mappo_algorithm.py:93-131,189-262; mappo_trainer.py:101-175) on the HIP
kernels, replayed against the golden episode + update recorded from the imported reference MAPPOTrainer(seed=0)
(tests/golden/mappo_seed0_T200.npz): get_actions, get_values, obs_dict_to_state, _format_env_actions, _store_transition,
RolloutBuffer, list-form compute_gae and update() fed through buffer.add."""
import numpy as np
import pytest
import torch

from tests.golden_util import assert_losses_close
from tests.test_oracle_mappo import load

pytestmark = pytest.mark.gpu


def _trainer(g):
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    cfg = dict(ENV_CONFIG, episode_length=int(g["episode_length"]))
    return MAPPOTrainer(cfg, seed=int(g["seed"]), device="cuda:0", n_envs=1, verbose=False)


def _obs_dict(g, t):
    """The reference's observation dict of step t: keys in arrival (insertion) order."""
    rk = g["step_rank"][t]
    return {f"node_{int(np.where(rk == r)[0][0])}": g["step_obs"][t][int(np.where(rk == r)[0][0])] for r in range(int((rk >= 0).sum()))}


def test_dict_api_replays_the_reference_episode_and_update():
    import mappo_algorithm as ma
    g = load()
    tr = _trainer(g)
    m = tr.mappo
    assert isinstance(ma.RolloutBuffer(), ma.RolloutBuffer) and len(ma.RolloutBuffer()) == 0
    T = int(g["episode_length"])
    rew = np.where(g["step_emitted"], g["step_reward"], 0.0)
    i = 0
    for t in range(T):
        obs_dict = _obs_dict(g, t)
        k = len(obs_dict)
        state = tr.obs_dict_to_state(obs_dict)
        assert state.dtype == np.float32 and np.array_equal(state, g["step_obs"][t].reshape(-1))
        actions, logps = m.get_actions(obs_dict, explore=True, noise=g["noise"][i:i + k])
        assert list(actions) == list(obs_dict)
        env_actions = tr._format_env_actions(actions)
        for j, a in enumerate(obs_dict):
            np.testing.assert_allclose(actions[a], g["buf_actions"][i + j], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(logps[a], g["buf_logp"][i + j], rtol=1e-5, atol=1e-5)
            ag = int(a.split("_")[1])
            assert env_actions[a]["placement"] == int(g["env_placement"][t][ag])
            assert env_actions[a]["resource"].dtype == np.float32 and env_actions[a]["resource"].shape == (1,)
            np.testing.assert_allclose(env_actions[a]["resource"][0], g["env_resource"][t][ag], rtol=1e-6)
        value = m.get_values(state) if k else 0.0
        if k:
            np.testing.assert_allclose(value, g["buf_values"][i], rtol=1e-5, atol=1e-5)
        tr._store_transition(obs_dict, actions, logps, state, float(rew[t]), {"__all__": t == T - 1}, value)
        i += k
    assert len(m.buffer) == len(g["buf_rewards"]) == i
    data = m.buffer.get()
    assert set(data) == {"observations", "states", "actions", "rewards", "dones", "log_probs", "values"}
    assert np.array_equal(np.asarray(data["rewards"]), g["buf_rewards"]) and np.array_equal(np.asarray(data["dones"]), g["buf_dones"])
    # list-form compute_gae (mappo_algorithm.py:241-262) on the REFERENCE's own buffer columns
    adv, ret = m.compute_gae(list(g["buf_rewards"]), list(g["buf_values"]), list(g["buf_dones"]), 0)
    assert isinstance(adv, list) and len(adv) == i
    assert np.array_equal(np.float32(adv), np.float32(g["gae_adv"])) and np.array_equal(np.float32(ret), np.float32(g["gae_ret"]))
    # update() consumes what buffer.add collected (mappo_algorithm.py:264-345), 4 epochs
    out = m.update()
    assert len(m.buffer) == 0
    assert_losses_close([out["actor_loss"], out["critic_loss"], out["entropy"]], g["loss4"], "dict-API update x4")
    got = {f"actor.{k}": v.detach().cpu().numpy() for k, v in m.actor.state_dict().items()}
    got.update({f"critic.{k}": v.detach().cpu().numpy() for k, v in m.critic.state_dict().items()})
    for k, v in g["p4"].items():
        np.testing.assert_allclose(got[k], v, rtol=1e-4, atol=2e-6, err_msg=k)


def test_compute_gae_known_answer_and_next_value():
    """The survey's known answer (imported reference): r=[1,2,3], v=[.5,.4,.3], done=[F,F,T]."""
    g = load()
    m = _trainer(g).mappo
    adv, ret = m.compute_gae([1.0, 2.0, 3.0], [0.5, 0.4, 0.3], [False, False, True], 0)
    assert np.array_equal(np.float32(adv), np.float32([5.068387175000001, 4.436350000000001, 2.7]))
    assert np.array_equal(np.float32(ret), np.float32([5.568387175000001, 4.836350000000001, 3.0]))
    # an open list bootstraps from next_value: delta_last = r + gamma * next_value - v
    adv2, _ = m.compute_gae([1.0], [0.5], [False], 2.0)
    assert np.float32(adv2[0]) == np.float32(1.0 + 0.99 * 2.0 - 0.5)
    assert m.compute_gae([], [], [], 0) == ([], [])


def test_eval_actions_and_seeded_sampling_and_normalizer():
    g = load()
    tr = _trainer(g)
    m = tr.mappo
    obs_dict = _obs_dict(g, 0)
    acts, lps = m.get_actions(obs_dict, explore=False)
    x = torch.as_tensor(np.stack(list(obs_dict.values()))).cuda()
    mu = m.actor(x).cpu().numpy()
    for j, a in enumerate(obs_dict):
        assert np.array_equal(acts[a], mu[j]) and lps[a] == 0.0                     # mappo_algorithm.py:214-218
    # sampling draws eps exactly as Normal.rsample() does on the CPU generator: one [1, A] draw per agent, in dict order
    torch.manual_seed(123)
    a1, l1 = m.get_actions(obs_dict, explore=True)
    torch.manual_seed(123)
    eps = np.concatenate([torch.normal(torch.zeros(1, 11), torch.ones(1, 11)).numpy() for _ in obs_dict])
    a2, l2 = m.get_actions(obs_dict, explore=True, noise=eps)
    for a in obs_dict:
        assert np.array_equal(a1[a], a2[a]) and l1[a] == l2[a]
    # MAPPOActor.get_action_log_prob(obs, action): log-prob and entropy as torch.distributions.Normal gives them
    act = torch.as_tensor(np.stack([a1[a] for a in obs_dict])).cuda()
    _, lp, ent = m.actor.get_action_log_prob(x, act)
    dist = torch.distributions.Normal(torch.as_tensor(mu), torch.exp(m.actor.log_std.detach().cpu()).expand(len(obs_dict), 11))
    np.testing.assert_allclose(lp.cpu().numpy(), dist.log_prob(act.cpu()).sum(-1).numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ent.cpu().numpy(), dist.entropy().sum(-1).numpy(), rtol=1e-6)
    vn = m.value_normalizer
    v = torch.tensor([[2.0]], device="cuda")
    assert torch.allclose(vn.denormalize(vn(v)), v) and float(vn(v)) == pytest.approx(2.0 / np.sqrt(1 + 1e-8))
    assert tr._format_env_actions({"node_0": np.zeros(11, np.float32)})["node_0"]["placement"] == 0
    short = tr._format_env_actions({"node_2": np.array([0.0, 3.0, 1.0], np.float32)})["node_2"]       # zero-padded
    assert short["placement"] == 1 and short["resource"][0] == np.float32(0.1 + 0.9 * 0.5)
