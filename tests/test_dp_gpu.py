"""Two ranks (gloo over CUDA tensors, both on the box's single GPU) against one process on the union of the
environments: same Philox streams per global environment, so trajectories are identical and the parameters
after k updates agree to fp32 summation-order noise."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

T, B_TOTAL, UPDATES = 60, 12, 2


def _cfg():
    from marl_amd.config import ENV_CONFIG
    cfg = dict(ENV_CONFIG)
    cfg["episode_length"] = T
    return cfg


def _run(trainer):
    for _ in range(UPDATES):
        trainer.train_episode(explore=True, record_buffer=True)
        trainer.mappo.update()
    return trainer.mappo.flat_params.detach().cpu().numpy().copy()


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=B_TOTAL // world, rank=rank, world_size=world, verbose=False)
    np.save(os.path.join(out_dir, f"params_{rank}.npy"), _run(tr))
    dist.destroy_process_group()


def test_two_ranks_match_one_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    p0, p1 = (np.load(tmp_path / f"params_{r}.npy") for r in range(2))
    assert np.array_equal(p0, p1)
    from marl_amd.trainer import MAPPOTrainer
    single = _run(MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=B_TOTAL, verbose=False))
    np.testing.assert_allclose(p0, single, rtol=2e-4, atol=2e-6)


# ---- a rank whose shard records nothing must still join every collective (no deadlock, no early return) ------------------
def _empty_cfg():
    cfg = _cfg()
    cfg.update(task_arrival_rate=1e-9, task_arrival_rate_range=(1e-9, 1e-9))      # Poisson(1e-9 dt): never an arrival
    return cfg


def _empty_shard_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg() if rank == 0 else _empty_cfg(), seed=3, device="cuda:0", n_envs=6, rank=0, world_size=world,
                      verbose=False)
    tr.train_episode(explore=True, record_buffer=True)
    e_local = tr.mappo.buffer.totals()[0]
    out = tr.mappo.update()
    np.save(os.path.join(out_dir, f"empty_{rank}.npy"), tr.mappo.flat_params.detach().cpu().numpy())
    np.save(os.path.join(out_dir, f"empty_meta_{rank}.npy"), np.array([e_local, tr.mappo.last_update["transitions"],
                                                                      out["critic_loss"]]))
    dist.destroy_process_group()


def test_rank_with_empty_shard_joins_the_collectives(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_empty_shard_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)     # a deadlock would hang here
    p0, p1 = (np.load(tmp_path / f"empty_{r}.npy") for r in range(2))
    m0, m1 = (np.load(tmp_path / f"empty_meta_{r}.npy") for r in range(2))
    assert m1[0] == 0 and m0[0] > 0 and m0[1] == m1[1] == m0[0]      # rank 1 recorded nothing; both saw the global count
    assert np.array_equal(p0, p1)                                     # replicas stay identical
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=6, verbose=False)
    tr.train_episode(explore=True, record_buffer=True)
    out = tr.mappo.update()
    assert np.array_equal(p0, tr.mappo.flat_params.detach().cpu().numpy())    # zeros added to the sums change nothing
    assert out["critic_loss"] == m0[2]


def _overflow_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=6, rank=rank, world_size=world, verbose=False,
                      capacity_entries=8 if rank == 1 else None)          # rank 1's storage overflows in the first steps
    raised = ""
    try:
        tr.train_episode(explore=True, record_buffer=True)
        tr.mappo.update()
    except RuntimeError as e:
        raised = str(e)
    open(os.path.join(out_dir, f"overflow_{rank}.txt"), "w").write(raised)
    dist.destroy_process_group()


def test_overflow_on_one_rank_raises_on_all(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_overflow_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = (open(tmp_path / f"overflow_{r}.txt").read() for r in range(2))
    assert "another rank" in m0 and "capacity 8 entries" in m1


# ---- the production backend on one rank: RCCL all-reduce ordered against this library's kernels ---------------------------
def _rccl_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      EDGEMARL_SYNC_ALWAYS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=B_TOTAL, rank=0, world_size=1, verbose=False)
    assert tr.mappo.sync.active and not tr.mappo.sync._drain
    np.save(os.path.join(out_dir, "rccl.npy"), _run(tr))
    dist.destroy_process_group()


def test_rccl_backend_one_rank_equals_no_collective(tmp_path):
    """backend "nccl" (= RCCL) with the collectives forced on in a one-rank group: every all_reduce of MAPPO.update runs
    on RCCL's stream, ordered by torch against the kernels this library queued on the current stream through raw stream
    handles.  A sum over one rank must leave every bit as it was."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    from marl_amd.trainer import MAPPOTrainer
    single = _run(MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=B_TOTAL, verbose=False))
    assert np.array_equal(np.load(tmp_path / "rccl.npy"), single)


# ---- independent PPO on the scaled topology (marl_amd/ippo.py) -------------------------------------------------------
def _scaled_cfg():
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG)
    cfg.update(n_nodes=8, n_deciders=24, node_cpu_capacity=np.linspace(1.0, 3.0, 8) * 1e9, episode_length=40)
    return cfg


def _run_ippo(trainer):
    for _ in range(UPDATES):
        trainer.train_episode(explore=True, record_buffer=True)
        trainer.algo.update()
    return np.concatenate([trainer.algo.actor.flat.cpu().numpy().ravel(), trainer.algo.critic.flat.cpu().numpy().ravel()])


def _ippo_worker(rank, world, port, out_dir, algorithm="ippo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.ippo import IPPOTrainer
    tr = IPPOTrainer(_scaled_cfg(), seed=5, device="cuda:0", n_envs=B_TOTAL // world, rank=rank, world_size=world, verbose=False,
                     algorithm=algorithm)
    np.save(os.path.join(out_dir, f"ippo_{rank}.npy"), _run_ippo(tr))
    dist.destroy_process_group()


@pytest.mark.parametrize("algorithm", ["ippo", "mappo"])
def test_ippo_two_ranks_match_one_process(tmp_path, algorithm):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_ippo_worker, args=(2, port, str(tmp_path), algorithm), nprocs=2, join=True)
    p0, p1 = (np.load(tmp_path / f"ippo_{r}.npy") for r in range(2))
    assert np.array_equal(p0, p1)
    from marl_amd.ippo import IPPOTrainer
    single = _run_ippo(IPPOTrainer(_scaled_cfg(), seed=5, device="cuda:0", n_envs=B_TOTAL, verbose=False, algorithm=algorithm))
    d = np.abs(p0 - single)
    # Adam turns a last-bit difference of a ~1e-8 gradient into a fraction of lr: bound the outliers (tests/test_ippo_gpu.py)
    assert (d > 2e-6 + 2e-4 * np.abs(single)).mean() < 1e-4 and d.max() <= 3e-4 * 4 * UPDATES


# ---- the constructor's reset probe stops at the same seed + attempt on every rank (mappo_trainer.py:53-58) ---------------
def _sparse_cfg():
    cfg = _cfg()
    cfg.update(task_arrival_rate=2, task_arrival_rate_range=(2, 2))      # reset: Poisson(0.2) arrivals, empty 82 % of the time
    return cfg


def _probe_worker(rank, world, port, out_dir, seed):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_sparse_cfg(), seed=seed, device="cuda:0", n_envs=1, rank=rank, world_size=world, verbose=False)
    np.save(os.path.join(out_dir, f"probe_{rank}.npy"), np.array([tr.reset_attempts]))
    np.save(os.path.join(out_dir, f"probe_obs_{rank}.npy"), tr.env.obs.cpu().numpy())
    dist.destroy_process_group()


def test_reset_probe_is_rank_consistent(tmp_path):
    """One environment per rank, arrivals rare: find a seed where, at the attempt the one-process run on the union stops at,
    rank 0's environment is EMPTY and rank 1's is not.  A rank-local test would send rank 0 on to the next seed; the
    all-reduced test stops both where the union stops, with the union's observations."""
    from marl_amd.env import BatchedEdgeSimGym
    from marl_amd.trainer import MAPPOTrainer
    env = BatchedEdgeSimGym(_sparse_cfg(), n_envs=2, device="cuda:0", seed=0)
    seed = None
    for s in range(200):
        _, active = env.reset(seed=s)
        a = (active != 0).cpu().numpy()
        if not a[0] and a[1]:
            seed = s
            break
    env.close()
    assert seed is not None
    union = MAPPOTrainer(_sparse_cfg(), seed=seed, device="cuda:0", n_envs=2, verbose=False)
    assert union.reset_attempts == 1
    mp.spawn(_probe_worker, args=(2, _free_port(), str(tmp_path), seed), nprocs=2, join=True)
    a0, a1 = (int(np.load(tmp_path / f"probe_{r}.npy")[0]) for r in range(2))
    assert a0 == a1 == union.reset_attempts
    want = union.env.obs.cpu().numpy()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"probe_obs_{r}.npy")[0], want[r])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


# ---- an overflowed batch is followed by a fitting one (the error flag is per batch, not per storage) -------------------------
def test_update_recovers_after_an_overflowed_batch():
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=6, verbose=False, capacity_entries=200)
    tr.train_episode(explore=True, record_buffer=True)           # ~6 x 60 x 0.87 = 310 entries: does not fit 200
    with pytest.raises(RuntimeError, match="capacity 200 entries"):
        tr.mappo.update()
    before = tr.mappo.flat_params.clone()
    # a batch that fits (low load) on a storage whose flag an earlier overflowed batch left set
    low = MAPPOTrainer({**_cfg(), "task_arrival_rate": 2, "task_arrival_rate_range": (2, 2)}, seed=3, device="cuda:0", n_envs=6,
                       verbose=False, capacity_entries=200)
    low.mappo.buffer.t["err_flag"].fill_(8)                      # what an earlier overflowed batch left behind
    low.train_episode(explore=True, record_buffer=True)
    out = low.mappo.update()
    assert np.isfinite(out["critic_loss"]) and low.mappo.last_update["transitions"] > 0
    # and on the trainer that really overflowed: the next update() neither raises "capacity exceeded" on an empty storage nor
    # changes anything
    assert tr.mappo.update() == {}
    assert torch.equal(before, tr.mappo.flat_params)


def test_update_refuses_list_transitions_beside_a_recorded_episode():
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(_cfg(), seed=3, device="cuda:0", n_envs=2, verbose=False)
    tr.train_episode(explore=True, record_buffer=True)
    z = np.zeros
    tr.mappo.buffer.add(obs=z(14, np.float32), state=z(140, np.float32), action=z(11, np.float32), reward=0.0, done=False,
                        log_prob=0.0, value=0.0)
    with pytest.raises(RuntimeError, match="buffer.add"):
        tr.mappo.update()


def test_bench_collective_messages_are_the_trainers_buffers():
    import bench
    from marl_amd.ippo import IPPOTrainer
    from marl_amd.trainer import MAPPOTrainer
    m = dict(bench.COLLECTIVE_MESSAGES)
    tr = MAPPOTrainer(_cfg(), seed=0, device="cuda:0", n_envs=2, verbose=False)
    assert tuple(tr.mappo.flat_grads.shape) == m["mappo_grads"]
    from marl_amd.config import SCALED_ENV_CONFIG
    cfg = dict(SCALED_ENV_CONFIG, episode_length=4)
    for alg, key in (("ippo", "ippo_grads"), ("mappo", "mappo_scaled_grads")):
        t2 = IPPOTrainer(cfg, seed=0, device="cuda:0", n_envs=2, verbose=False, algorithm=alg)
        assert tuple(t2.algo.grads.shape) == m[key]
        t2.env.close()
