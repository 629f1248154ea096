"""CPU: oracle/edge_marl_oracle.py pinned against episodes recorded from the reference's EdgeMARLEnv
(pymarl/src/envs/edge_marl_env.py) - bit-exact on every output the reference returns."""
import numpy as np
import pytest

from oracle import edge_marl_oracle as mo
from tests import marl_env_fixture as fx

SUMMARY_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "deadline_violation_rate", "load_balance_jain")


@pytest.mark.parametrize("name", fx.FIXTURES)
def test_replay_matches_reference_bit_exact(name):
    args, z, out, tape = fx.replay(name)
    assert np.array_equal(out["reward"], z["reward"])
    assert np.array_equal(out["terminated"].astype(np.uint8), z["terminated"])
    assert np.array_equal(out["info"], z["info"])
    assert np.array_equal(out["obs"], z["obs"])
    assert np.array_equal(z["state0"], z["obs"][0].reshape(-1))          # get_state = concat of the obs rows
    assert [out["summary"][k] for k in SUMMARY_KEYS] == list(z["summary"])
    n = int(args["n_nodes"])
    assert list(z["sizes"]) == [n + 4, n * (n + 4), n, 1 if args["action_mode"] == "discrete" else n + 1]
    assert z["terminated"][-1] == 1 and z["terminated"][:-1].sum() == 0


@pytest.mark.parametrize("name", fx.FIXTURES)
def test_generator_stream_matches_reference(name):
    """numpy's default_rng(seed) inside the restatement reproduces the recorded episode without the log
    (constructor reset + explicit reset, as the recording did)."""
    args, z, _ = fx.load(name)
    env = mo.EdgeMARLOracle(dict(args))
    env.reset()
    assert np.array_equal(env.obs(), z["obs"][0])
    for t in range(len(z["reward"])):
        r, term, _ = env.step(z["actions"][t])
        assert r == z["reward"][t] and term == bool(z["terminated"][t])
    assert np.array_equal(env.obs(), z["obs"][-1])


@pytest.mark.parametrize("name", fx.FIXTURES)
def test_structured_tape_round_trip(name):
    args, z, out, tape = fx.replay(name)
    env = mo.EdgeMARLOracle.__new__(mo.EdgeMARLOracle)
    mo.EdgeMARLOracle.__init__(env, dict(args), draws=mo.TapeDraws(tape))
    assert np.array_equal(env.obs(), z["obs"][0])
    for t in range(len(z["reward"])):
        r, _, info = env.step(z["actions"][t])
        assert r == z["reward"][t]
    assert np.array_equal(env.obs(), z["obs"][-1])
    if args["task_arrival_mode"] != "random":
        assert np.isnan(tape["rate_u"]).all()


def test_hybrid_action_conversion_edges():
    n = 5
    assert mo.convert_hybrid_action(np.zeros(n + 1), n) == (0, 0.55)
    p, s = mo.convert_hybrid_action(np.array([0, 0, 0, 2.0, 0, -1.0]), n)
    assert (p, s) == (3, 0.1)
    p, s = mo.convert_hybrid_action(np.array([0.5, 0.7]), n)              # short vectors are zero-padded
    assert p == 1 and s == 0.55
    p, s = mo.convert_hybrid_action(np.array([1e-9, 5e-9, 0, 0, 0, 9.0]), n)
    assert (p, s) == (0, 1.0)


def test_philox_mode_is_deterministic_and_sane():
    args = mo.default_args()
    args["episode_length"] = 30
    a = mo.EdgeMARLOracle(dict(args), draws=mo.PhiloxDraws(99, 3))
    b = mo.EdgeMARLOracle(dict(args), draws=mo.PhiloxDraws(99, 3))
    c = mo.EdgeMARLOracle(dict(args), draws=mo.PhiloxDraws(99, 4))
    acts = np.arange(10)[::-1]
    ra = [a.step(acts)[0] for _ in range(30)]
    rb = [b.step(acts)[0] for _ in range(30)]
    rc = [c.step(acts)[0] for _ in range(30)]
    assert ra == rb and ra != rc
    assert all(np.isfinite(ra)) and a.step_no == 30
    o = a.obs()
    assert (o[:, 2] >= 0.5).all() and (o[:, 2] <= 5.0).all() and (o[:, 3] >= 1.0).all() and (o[:, 3] <= 10.0).all()


@pytest.mark.parametrize("name", ("cyclic_discrete", "fixed_small"))
def test_host_episode_summary_matches_reference(name):
    """marl_amd.marl_env.episode_summary_from_log (host side of get_episode_summary) on the per-step rows the kernel logs."""
    from marl_amd.marl_env import episode_summary_from_log
    args, z, out, _ = fx.replay(name)
    row = np.concatenate([out["info"], out["info"][:, :1] * 0.0], axis=1)
    env = mo.EdgeMARLOracle(dict(args))
    env.reset()
    lat = []
    for t in range(len(z["reward"])):
        env.step(z["actions"][t])
        lat.append(env.metrics["latency"][-1])
    row[:, 5] = lat
    log = np.stack([row, row[::-1]], axis=1)                                  # two environments; the second reversed in time
    got = episode_summary_from_log(log)
    for j, k in enumerate(SUMMARY_KEYS):
        assert got[k][0] == z["summary"][j], k
    assert got["p99_latency_ms"][1] == got["p99_latency_ms"][0]
    assert episode_summary_from_log(log[:0])["avg_energy_J"].tolist() == [0.0, 0.0]
