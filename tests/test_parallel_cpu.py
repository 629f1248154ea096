"""world_size-2 gloo test (CPU) of the data-parallel exchange of the MAPPO update: summing each rank's
gradient-of-sums (scaled by 1/L_global, advantages normalised with the all-reduced moments) reproduces the
single-process full-batch gradient of the torch-CPU restatement."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from oracle import mappo_oracle as mo
from tests.test_oracle_mappo import flatten_buffer, load


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_grads(p, batch, lo, hi, mean, std, n_global):
    """gradient of  sum_over_local_rows(loss terms) / n_global  for rows [lo, hi)."""
    pt = {k: v.clone().requires_grad_(True) for k, v in mo.to_torch(p).items()}
    sl = slice(lo, hi)
    obs, st = torch.as_tensor(batch["obs"][sl]), torch.as_tensor(batch["states"][sl])
    act = torch.as_tensor(batch["actions"][sl])
    old = torch.as_tensor(batch["old_logp"][sl], dtype=torch.float32)
    adv = (torch.as_tensor(batch["adv"][sl], dtype=torch.float32) - mean) / (std + 1e-8)
    ret = torch.as_tensor(batch["returns"][sl], dtype=torch.float32)
    mu = mo.actor_mu(pt, obs)
    logp, ent = mo.gaussian_logp_entropy(mu, pt["actor.log_std"], act)
    ratio = torch.exp(logp - old)
    obj = torch.min(ratio * adv, torch.clamp(ratio, 0.8, 1.2) * adv).sum()
    v = mo.critic_v(pt, st)
    loss = (-obj + 0.5 * 0.5 * ((v - ret) ** 2).sum() - 0.01 * ent.sum()) / n_global
    loss.backward()
    return torch.cat([pt[k].grad.reshape(-1) for k in mo.ACTOR_KEYS + mo.CRITIC_KEYS])


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.parallel import GradSync, shard_range
    g = load()
    obs, states, _ = flatten_buffer(g)
    batch = {"obs": obs, "states": states, "actions": g["buf_actions"], "old_logp": g["buf_logp"],
             "adv": g["gae_adv"], "returns": g["gae_ret"]}
    L = len(obs)
    lo, hi = shard_range(L, rank, world)
    sync = GradSync()
    a32 = np.float32(g["gae_adv"][lo:hi]).astype(np.float64)
    moments = torch.tensor([float(hi - lo), a32.sum(), (a32 * a32).sum()], dtype=torch.float64)
    n_global = sync.sum_moments(moments)
    assert n_global == L
    n, s1, s2 = moments.tolist()
    mean = np.float32(s1 / n)
    std = np.float32(np.sqrt((s2 - s1 * s1 / n) / (n - 1)))
    flat = _rank_grads(g["init"], batch, lo, hi, float(mean), float(std), n_global)
    sync.sum_grads(flat)
    params = torch.arange(10, dtype=torch.float32) * (rank + 1)
    sync.broadcast_params(params)
    assert torch.equal(params, torch.arange(10, dtype=torch.float32))
    np.save(os.path.join(out_dir, f"grads_{rank}.npy"), flat.numpy())
    dist.destroy_process_group()


def test_two_rank_gradient_of_sums_equals_full_batch(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g0, g1 = (np.load(tmp_path / f"grads_{r}.npy") for r in range(world))
    assert np.array_equal(g0, g1)          # replicas hold identical reduced gradients
    g = load()
    obs, states, _ = flatten_buffer(g)
    batch = {"obs": obs, "states": states, "actions": g["buf_actions"], "old_logp": g["buf_logp"],
             "adv": g["gae_adv"], "returns": g["gae_ret"]}
    _, _, grads0, _, _ = mo.update(g["init"], batch, n_iter=1)
    want = np.concatenate([grads0[k].reshape(-1) for k in mo.ACTOR_KEYS + mo.CRITIC_KEYS])
    np.testing.assert_allclose(g0, want, rtol=2e-4, atol=1e-6)


def test_shard_range_partitions():
    from marl_amd.parallel import shard_range
    for n, w in ((10, 3), (8, 8), (65536, 8), (5, 2)):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _counts_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.parallel import GradSync
    got = GradSync().sum_counts([100, 0, 0] if rank == 0 else [100, 7, 8], torch.device("cpu"))
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), np.array(got))
    dist.destroy_process_group()


def test_sum_counts_gives_every_rank_the_global_totals(tmp_path):
    """The gate of MAPPO.update / IPPO.update: a rank with an empty shard (0 entries) or an overflow flag elsewhere sees the
    same totals as its peers, so all ranks take the same branch before the first gradient collective."""
    mp.spawn(_counts_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    c0, c1 = (np.load(tmp_path / f"counts_{r}.npy") for r in range(2))
    assert c0.tolist() == c1.tolist() == [200, 7, 8]
    from marl_amd.parallel import GradSync
    assert GradSync(enabled=False).sum_counts([3, 0], torch.device("cpu")) == [3, 0]      # single process: identity


def test_bench_launcher_builds_the_rank_command():
    """`python bench.py --gpus N` (no WORLD_SIZE in the environment) starts N ranks as children through
    torch.distributed.run with the same argv; the parent makes no GPU call."""
    import sys

    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "50", "--warmup", "5"], 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    script = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[script + 1:] == ["--gpus", "8", "--steps", "50", "--warmup", "5"]
    assert 1024 < bench.free_port() < 65536


def test_bench_parent_launches_children_and_passes_exit_code(monkeypatch):
    import bench
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env)
        return 7

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr("sys.argv", ["bench.py", "--gpus", "2", "--no-cpu-baseline"])
    try:
        bench.main()
        raise AssertionError("the parent must exit with the children's code")
    except SystemExit as e:
        assert e.code == 7
    assert "--nproc-per-node=2" in seen["cmd"] and seen["cmd"][-3:] == ["--gpus", "2", "--no-cpu-baseline"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
