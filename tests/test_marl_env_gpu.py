"""GPU: the HIP EdgeMARLEnv (csrc/edge_marl_env.hip through the C ABI) against the reference's recorded episodes (tape mode,
bit-exact) and against oracle/edge_marl_oracle.py on the in-kernel Philox stream (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import edge_marl_oracle as mo
from tests import marl_env_fixture as fx

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SUMMARY_KEYS = ("avg_latency_ms", "p99_latency_ms", "avg_energy_J", "throughput_tps", "deadline_violation_rate", "load_balance_jain")


def _tape_row(env_mod, tape, s, B, reset=False):
    rep = lambda a: np.repeat(np.asarray(a)[None], B, axis=0)      # noqa: E731
    arrays = {"task_workload": rep(tape["task_wl"][s]), "task_data": rep(tape["task_data"][s]),
              "task_deadline_offset": rep(tape["task_ddl_off"][s])}
    if not reset:
        arrays.update({"rate_u": rep(tape["rate_u"][s]), "dt_raw": rep(tape["dt_raw"][s]), "bw_u": rep(tape["bw_u"][s]),
                       "net_lat": rep(tape["lat"][s])})
    return env_mod.MarlDeviceTape(arrays, DEV)


@pytest.mark.parametrize("lanes", (1, 16))
@pytest.mark.parametrize("name", fx.FIXTURES)
def test_reference_episode_bit_exact(name, lanes):
    from marl_amd import marl_env
    args, z, out, tape = fx.replay(name)
    if lanes == 16 and int(args["n_nodes"]) > 16:
        pytest.skip("16 lanes per environment need n_nodes <= 16")
    B, steps = 3 if lanes == 1 else 21, len(z["reward"])
    kw = {k: v for k, v in args.items() if k != "seed"}
    env = marl_env.EdgeMARLEnv(n_envs=B, device=DEV, **kw)
    env.set_lanes_per_env(lanes)
    env.enable_debug_outputs()
    env.reset(tape=_tape_row(marl_env, tape, 0, B, reset=True))
    assert np.array_equal(env.get_obs().cpu().numpy(), np.repeat(z["obs"][0][None], B, 0))
    assert np.array_equal(env.get_state().cpu().numpy()[0], z["state0"])
    acts = torch.as_tensor(z["actions"], device=DEV)
    for t in range(steps):
        a = acts[t][None].expand(B, *acts[t].shape).contiguous()
        r, term, info = env.step(a, tape=_tape_row(marl_env, tape, t + 1, B))
        assert np.array_equal(r.cpu().numpy(), np.full(B, z["reward"][t])), f"reward, step {t}"
        assert np.array_equal(term.cpu().numpy(), np.full(B, bool(z["terminated"][t])))
        got = np.stack([info[k].cpu().numpy() for k in marl_env.INFO_KEYS], axis=1)
        assert np.array_equal(got, np.repeat(z["info"][t][None], B, 0)), f"info, step {t}"
        assert np.array_equal(env.get_obs().cpu().numpy(), np.repeat(z["obs"][t + 1][None], B, 0)), f"obs, step {t}"
    summ = env.get_episode_summary()
    for j, k in enumerate(SUMMARY_KEYS):
        assert np.array_equal(summ[k], np.full(B, z["summary"][j])), k
    env.close()


def _philox_case(kw, B, steps, seed, env_offset, rng, lanes=1):
    from marl_amd import marl_env
    env = marl_env.EdgeMARLEnv(n_envs=B, device=DEV, env_offset=env_offset, seed=seed, **kw)
    env.set_lanes_per_env(lanes)
    env.enable_debug_outputs()
    n = env.n_agents
    args = dict(marl_env.MARL_ENV_ARGS); args.update(kw)
    oracles = [mo.EdgeMARLOracle(dict(args), draws=mo.PhiloxDraws(seed, env_offset + b, episode=1)) for b in range(B)]
    assert np.array_equal(env.get_obs().cpu().numpy(), np.stack([o.obs() for o in oracles]))
    hybrid = args["action_mode"] == "hybrid"
    for t in range(steps):
        if hybrid:
            a = rng.uniform(-1.2, 1.2, size=(B, n, n + 1)).astype(np.float32)
            a[rng.rand(B) < 0.2, :, :n] = 0.0
        else:
            a = rng.randint(-1, n + 1, size=(B, n)).astype(np.int32)
        r, term, info = env.step(torch.as_tensor(a, device=DEV))
        exp = [o.step(a[b]) for b, o in enumerate(oracles)]
        assert np.array_equal(r.cpu().numpy(), np.array([e[0] for e in exp])), f"reward, step {t}"
        assert np.array_equal(term.cpu().numpy(), np.array([e[1] for e in exp]))
        for k in marl_env.INFO_KEYS:
            assert np.array_equal(info[k].cpu().numpy(), np.array([e[2][k] for e in exp])), f"{k}, step {t}"
        assert np.array_equal(env.latency.cpu().numpy(), np.array([o.last["latencies"] for o in oracles]))
        assert np.array_equal(env.get_obs().cpu().numpy(), np.stack([o.obs() for o in oracles])), f"obs, step {t}"
    st = env.raw_state()
    assert np.array_equal(st["load"], np.stack([o.load for o in oracles]))
    assert np.array_equal(st["queue"], np.stack([o.queue for o in oracles]))
    assert np.array_equal(st["sim_time"], np.array([o.sim_time for o in oracles]))
    assert (st["step"] == steps).all()
    summ = env.get_episode_summary()
    for k in SUMMARY_KEYS:
        assert np.array_equal(summ[k], np.array([o.episode_summary()[k] for o in oracles])), k
    env.close()


@pytest.mark.parametrize("lanes", (1, 16))
def test_philox_stream_cyclic_hybrid(lanes):
    _philox_case({"action_mode": "hybrid", "episode_length": 20}, B=40, steps=22, seed=2024, env_offset=1000,
                 rng=np.random.RandomState(0), lanes=lanes)


def test_philox_stream_random_rate_wide_two_workgroups():
    kw = {"n_nodes": 24, "node_cpu_capacity": list(np.linspace(1e9, 3e9, 24)), "task_arrival_mode": "random", "episode_length": 50}
    _philox_case(kw, B=150, steps=5, seed=7, env_offset=0, rng=np.random.RandomState(1))


@pytest.mark.parametrize("lanes", (1, 16))
def test_philox_stream_fixed_rate_tiny(lanes):
    kw = {"n_nodes": 3, "node_cpu_capacity": [1e9, 2e9, 3e9], "task_arrival_mode": "fixed", "task_arrival_rate": 30.0, "episode_length": 9}
    _philox_case(kw, B=300, steps=9, seed=5, env_offset=2**32 - 100, rng=np.random.RandomState(2), lanes=lanes)


def test_philox_stream_sixteen_nodes_random_rate():
    kw = {"n_nodes": 16, "node_cpu_capacity": list(np.linspace(1e9, 3e9, 16)), "task_arrival_mode": "random", "episode_length": 30}
    _philox_case(kw, B=35, steps=6, seed=21, env_offset=3, rng=np.random.RandomState(4), lanes=16)


def test_lane_variants_agree_at_full_size():
    """4,096 environments, 60 steps: the two kernel variants produce identical outputs and state."""
    from marl_amd import marl_env
    B = 4096
    envs = []
    for lanes in (1, 16):
        e = marl_env.EdgeMARLEnv(n_envs=B, device=DEV, seed=9, episode_length=60)
        e.set_lanes_per_env(lanes)
        e.seed(9); e.reset()
        envs.append(e)
    g = torch.Generator(device=DEV); g.manual_seed(5)
    for t in range(60):
        a = torch.randint(0, 10, (B, 10), device=DEV, dtype=torch.int32, generator=g)
        outs = [e.step(a) for e in envs]
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        for k in marl_env.INFO_KEYS:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
        assert torch.equal(envs[0].get_obs(), envs[1].get_obs())
    s0, s1 = envs[0].raw_state(), envs[1].raw_state()
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    with pytest.raises(RuntimeError):
        envs[0].set_lanes_per_env(4)


def test_philox_stream_max_nodes():
    kw = {"n_nodes": 64, "node_cpu_capacity": list(np.linspace(1e9, 3e9, 64)), "episode_length": 10}
    _philox_case(kw, B=70, steps=3, seed=11, env_offset=5, rng=np.random.RandomState(3))


def test_interface_and_seeding():
    from marl_amd import marl_env
    B = 17
    env = marl_env.EdgeMARLEnv(n_envs=B, device=DEV, seed=3, episode_length=4)
    n = env.n_agents
    assert env.get_env_info() == {"state_shape": n * (n + 4), "obs_shape": n + 4, "n_actions": n, "n_agents": n, "episode_limit": 4}
    assert env.get_obs_size() == 14 and env.get_state_size() == 140 and env.get_total_actions() == 10 and env.get_action_dim() == 1
    av = env.get_avail_actions()
    assert av.shape == (B, n, n) and av.dtype == torch.int32 and bool((av == 1).all())
    assert env.get_avail_agent_actions(2).shape == (B, n)
    assert torch.equal(env.get_obs_agent(4), env.get_obs()[:, 4])
    assert torch.equal(env.get_state(), env.get_obs().reshape(B, -1))
    with pytest.raises(ValueError):
        env.step(torch.zeros((B, n - 1), dtype=torch.int32, device=DEV))
    with pytest.raises(ValueError):
        marl_env.EdgeMARLEnv(n_envs=1, device=DEV, action_mode="continuous")
    acts = torch.arange(n, dtype=torch.int64, device=DEV).flip(0)[None].expand(B, n)      # int64 is converted
    first_obs = env.get_obs().clone()
    trace = []
    for t in range(4):
        r, term, info = env.step(acts)
        trace.append(r.clone())
        assert bool(term.all()) == (t == 3) and torch.equal(info["episode_limit"], term)
    assert len({float(x) for x in trace[0].cpu()}) == B                                    # environments draw independently
    env.reset()                                                                            # next episode: a different stream
    assert not torch.equal(env.get_obs(), first_obs)
    env.seed(3)
    env.reset()                                                                            # re-keyed: the first episode again
    assert torch.equal(env.get_obs(), first_obs)
    for t in range(4):
        r, _, _ = env.step(acts)
        assert torch.equal(r, trace[t])
    assert env.get_episode_summary()["avg_latency_ms"].shape == (B,)
    env.close()


def test_full_size_properties():
    """BASELINE-size batch: invariants that need no oracle (4,096 environments, a whole 300-step episode)."""
    from marl_amd import marl_env
    B, steps = 4096, 300
    env = marl_env.EdgeMARLEnv(n_envs=B, device=DEV, seed=1, episode_length=steps)
    n = env.n_agents
    g = torch.Generator(device=DEV); g.manual_seed(0)
    for t in range(steps):
        a = torch.randint(0, n, (B, n), device=DEV, dtype=torch.int32, generator=g)
        r, term, info = env.step(a)
        if t % 50 == 0 or t == steps - 1:
            assert bool(torch.isfinite(r).all()) and bool((r < 0).all())
            assert bool((info["throughput_tps"] >= n / 0.2 - 1e-9).all()) and bool((info["throughput_tps"] <= n / 0.02 + 1e-9).all())
            assert bool((info["load_balance_jain"] > 1.0 / n - 1e-12).all()) and bool((info["load_balance_jain"] <= 1.0 + 1e-12).all())
            v = info["deadline_violation_rate"] * n
            assert bool((v == v.round()).all())
            o = env.get_obs()
            assert bool(torch.isfinite(o).all()) and bool((o[:, :, 0] >= 0).all())
            # neighbour columns are the other agents' load ratios in index order
            assert torch.equal(o[:, 0, 5:], o[:, 1:, 0]) and torch.equal(o[:, n - 1, 5:], o[:, : n - 1, 0])
            assert bool(term.all()) == (t == steps - 1)
    st = env.raw_state()
    assert (st["step"] == steps).all() and (st["sim_time"] >= steps * 0.02 - 1e-9).all() and (st["sim_time"] <= steps * 0.2 + 1e-9).all()
    s = env.get_episode_summary()
    assert np.isfinite(s["p99_latency_ms"]).all() and (s["p99_latency_ms"] >= s["avg_latency_ms"] * 0.5).all()
    env.close()


def test_host_view_matches_batched_row_and_reference_types():
    """The n_envs = 1 adapter PyMARL's registry binds: host types, same numbers as row 0 of the batched environment."""
    from marl_amd import marl_env
    host = marl_env.EdgeMARLEnvHost(device=DEV, seed=4, episode_length=6)
    ref = marl_env.EdgeMARLEnv(n_envs=3, device=DEV, seed=4, episode_length=6)
    n = host.n_agents
    obs = host.get_obs()
    assert isinstance(obs, list) and len(obs) == n and obs[0].dtype == np.float32 and obs[0].shape == (n + 4,)
    assert np.array_equal(np.stack(obs), ref.get_obs()[0].cpu().numpy())
    assert np.array_equal(host.get_state(), np.concatenate(obs))
    assert host.get_avail_actions().shape == (n, n) and host.get_avail_agent_actions(0).dtype == np.int32
    rng = np.random.RandomState(0)
    for t in range(6):
        acts = rng.randint(-2, n + 2, size=n)
        feed = [torch.tensor(acts[i]) if i % 2 else np.int64(acts[i]) for i in range(n)]     # tensors and numpy scalars, :99-101
        r, term, info = host.step(feed)
        rb, tb, ib = ref.step(torch.as_tensor(np.repeat(acts[None], 3, 0).astype(np.int32), device=DEV))
        assert isinstance(r, float) and isinstance(term, bool) and isinstance(info["avg_latency_ms"], float)
        assert r == float(rb[0]) and term == bool(tb[0]) and info["episode_limit"] == term
        assert info["load_balance_jain"] == float(ib["load_balance_jain"][0])
    assert term
    s = host.get_episode_summary()
    assert s["p99_latency_ms"] == float(ref.get_episode_summary()["p99_latency_ms"][0])
    with pytest.raises(ValueError):
        host.step([0] * (n - 1))
    hy = marl_env.EdgeMARLEnvHost(device=DEV, seed=4, action_mode="hybrid")
    assert hy.get_action_dim() == n + 1
    r, _, _ = hy.step([np.array([0.1, 0.9])] * n)            # short vectors are zero padded: placement 1, resource norm 0.5
    hy2 = marl_env.EdgeMARLEnv(n_envs=1, device=DEV, seed=4, action_mode="hybrid")
    full = np.zeros((1, n, n + 1), np.float32); full[0, :, 0] = 0.1; full[0, :, 1] = 0.9
    assert r == float(hy2.step(torch.as_tensor(full, device=DEV))[0][0])
    host.close(); ref.close(); hy.close(); hy2.close()
