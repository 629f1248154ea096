"""Pins oracle/mappo_oracle.py (torch-CPU fp32 restatement) and the C GAE against the golden episode and
update recorded from the imported reference MAPPOTrainer(seed=0) (oracle/make_golden.py)."""
import os

import numpy as np
import torch

from oracle import c_oracle
from oracle import mappo_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mappo_seed0_T200.npz")
# configs[0] at its real length: the first episode and update of the reference's MAPPOTrainer(ENV_CONFIG, seed=0), plus one
# update with every attribute run_mappo_batch.py:255-261 overrides off its default (oracle/make_golden.py --mappo-full)
GOLD_FULL = os.path.join(os.path.dirname(GOLD), "mappo_seed0_T1000.npz")
FIXTURES = ("mappo_seed0_T200", "mappo_seed0_T1000")


def load(name="mappo_seed0_T200"):
    z = np.load(os.path.join(os.path.dirname(GOLD), name + ".npz"), allow_pickle=False)
    g = {k: z[k] for k in z.files}
    g["init"] = {k[5:]: g[k] for k in list(g) if k.startswith("init.")}
    g["p1"] = {k[3:]: g[k] for k in list(g) if k.startswith("p1.")}
    g["p4"] = {k[3:]: g[k] for k in list(g) if k.startswith("p4.")}
    g["ph"] = {k[3:]: g[k] for k in list(g) if k.startswith("ph.")}
    return g


def flatten_buffer(g):
    """Per-agent flattening of the steps in arrival order (mappo_trainer.py:160-175)."""
    T, n = int(g["episode_length"]), int(g["n_nodes"])
    obs, states, step_of = [], [], []
    for t in range(T):
        rk = g["step_rank"][t]
        for r in range(int((rk >= 0).sum())):
            a = int(np.where(rk == r)[0][0])
            obs.append(g["step_obs"][t][a])
            states.append(g["step_obs"][t].reshape(-1))
            step_of.append(t)
    return np.array(obs, np.float32), np.array(states, np.float32), np.array(step_of)


import pytest  # noqa: E402


@pytest.mark.parametrize("name", FIXTURES)
def test_rollout_quantities(name):
    g = load(name)
    p = mo.to_torch(g["init"])
    obs, states, step_of = flatten_buffer(g)
    a, logp, mu = mo.sample(p, torch.as_tensor(obs), torch.as_tensor(g["noise"]))
    np.testing.assert_allclose(a.numpy(), g["buf_actions"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(logp.numpy(), g["buf_logp"], rtol=1e-6, atol=1e-5)
    v = mo.rollout_value(p, torch.as_tensor(states))
    np.testing.assert_allclose(v.numpy(), g["buf_values"], rtol=1e-6, atol=1e-6)
    n = int(g["n_nodes"])
    i = 0
    for t in range(int(g["episode_length"])):
        rk = g["step_rank"][t]
        for r in range(int((rk >= 0).sum())):
            ag = int(np.where(rk == r)[0][0])
            pl, rs = mo.format_env_action(g["buf_actions"][i], n)
            assert pl == g["env_placement"][t][ag] and rs == g["env_resource"][t][ag]
            i += 1
    rew = np.where(g["step_emitted"], g["step_reward"], 0.0)
    assert np.array_equal(rew[step_of], g["buf_rewards"])
    assert float(rew.sum()) == float(g["episode_reward"]) or abs(rew.sum() - g["episode_reward"]) < 1e-9


@pytest.mark.parametrize("name", FIXTURES)
def test_gae_exact(name):
    g = load(name)
    adv, ret = mo.gae_flat(list(g["buf_rewards"]), list(g["buf_values"]), list(g["buf_dones"]))
    assert np.array_equal(adv, g["gae_adv"]) and np.array_equal(ret, g["gae_ret"])
    cadv, cret = c_oracle.gae(g["buf_rewards"], g["buf_values"], g["buf_dones"], 0.0, 0.99, 0.95)
    assert np.array_equal(cadv, g["gae_adv"]) and np.array_equal(cret, g["gae_ret"])


def test_survey_known_answers_of_config_0():
    """MAPPOTrainer(seed=0, cpu) first episode reward -1787.630, 883 transitions; first update() actor_loss
    -0.005183442062996146, critic_loss 1765.050537109375, entropy 15.607886552810669."""
    g = load("mappo_seed0_T1000")
    assert int(g["episode_length"]) == 1000 and len(g["buf_rewards"]) == 883
    assert abs(float(g["episode_reward"]) - (-1787.630)) < 5e-4
    assert g["loss4"].tolist() == [-0.005183442062996146, 1765.050537109375, 15.607886552810669]


def test_update_with_overridden_hyperparameters_matches_reference():
    """run_mappo_batch.py:255-261: distinct actor/critic learning rates, clip, entropy coefficient and 6 epochs."""
    g = load("mappo_seed0_T1000")
    obs, states, _ = flatten_buffer(g)
    batch = {"obs": obs, "states": states, "actions": g["buf_actions"], "old_logp": g["buf_logp"],
             "adv": g["gae_adv"], "returns": g["gae_ret"]}
    lr_a, lr_c, clip, ent, n_iter = g["hyper"].tolist()
    assert (lr_a, lr_c, clip, ent, n_iter) == (1e-4, 1e-3, 0.15, 0.005, 6.0)
    ph, losses, _, _, _ = mo.update(g["init"], batch, n_iter=int(n_iter), lr_actor=lr_a, lr_critic=lr_c, clip=clip, ecoef=ent)
    from tests.golden_util import assert_losses_close      # the actor loss is a cancelling mean: held to its summands' scale
    assert_losses_close(np.mean(np.array(losses), axis=0), g["lossh"], "overridden hyper-parameters", rel=1e-6)
    for k, v in g["ph"].items():
        np.testing.assert_allclose(ph[k], v, rtol=1e-5, atol=1e-7, err_msg=k)
    # and the overrides matter: the default update ends somewhere else
    assert not np.allclose(g["ph"]["critic.fc1.weight"], g["p4"]["critic.fc1.weight"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("name", FIXTURES)
def test_update_matches_reference(name):
    g = load(name)
    obs, states, _ = flatten_buffer(g)
    batch = {"obs": obs, "states": states, "actions": g["buf_actions"], "old_logp": g["buf_logp"],
             "adv": g["gae_adv"], "returns": g["gae_ret"]}
    p1, losses1, _, _, _ = mo.update(g["init"], batch, n_iter=1)
    from tests.golden_util import assert_losses_close      # the first epoch's actor loss is -mean(adv_norm) = 0 up to rounding
    assert_losses_close(losses1[0], g["loss1"], "1 epoch", rel=1e-6)
    for k, v in g["p1"].items():
        np.testing.assert_allclose(p1[k], v, rtol=1e-6, atol=1e-8, err_msg=k)
    p4, losses4, _, _, _ = mo.update(g["init"], batch, n_iter=4)
    assert_losses_close(np.mean(np.array(losses4), axis=0), g["loss4"], "4 epochs", rel=1e-6)
    for k, v in g["p4"].items():
        np.testing.assert_allclose(p4[k], v, rtol=1e-5, atol=1e-7, err_msg=k)


def test_restatement_on_a_shipped_reference_checkpoint():
    """The torch-CPU restatement reproduces what the imported reference networks computed from the reference's own trained
    checkpoint (tests/golden/ckpt_seed0.npz, written by oracle/make_golden_ckpt.py from mappo_seed_0.pt)."""
    g = np.load(os.path.join(os.path.dirname(GOLD), "ckpt_seed0.npz"), allow_pickle=False)
    p = mo.to_torch({k: g[k] for k in g.files if k.startswith(("actor.", "critic."))})
    mu = mo.actor_mu(p, torch.as_tensor(g["obs"]))
    np.testing.assert_allclose(mu.numpy(), g["mu"], rtol=1e-6, atol=1e-5)
    lp, ent = mo.gaussian_logp_entropy(mu, p["actor.log_std"], torch.as_tensor(g["actions"]))
    np.testing.assert_allclose(lp.numpy(), g["logp"], rtol=1e-6)
    np.testing.assert_allclose(ent.numpy(), g["entropy"], rtol=1e-6)
    np.testing.assert_allclose(mo.critic_v(p, torch.as_tensor(g["states"])).numpy(), g["values"], rtol=1e-6)
    assert float(g["adam_step"]) == 20000.0 and g["norm"].tolist() == [0.0, 1.0, 1.0]
