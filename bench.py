#!/usr/bin/env python3
"""bench.py - headline benchmark of the hot path (BASELINE.json: agent-steps/sec of the env step).

Workload at N=1 = BASELINE.json configs[1]: "EdgeSimGym default topology, 4096 batched envs, random
actions, step-only kernel on one MI355X".  One "step" = one em_step launch over the whole batch.
Inputs (action pool, environment state) are resident in HBM before the timed region.  With N>1
(torchrun, one rank per GPU) every rank steps its own 4096-environment shard: environments are
independent, so there is no data-path collective and scaling is weak.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs B] [--no-cpu-baseline] [--no-mappo]

Prints ONE JSON line on rank 0 with the `roofline` (dominant kernel, HBM) and `cpu_baseline` objects.
The second half of BASELINE.json's metric ("MAPPO samples/sec") rides along in the `mappo` object:
configs[2]/[3], 65,536 environments per GPU, one iteration = 1000-step rollout + GAE + 4 PPO epochs, data
parallel over the ranks (one fused gradient all-reduce per epoch on RCCL); it is not part of `value`.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# dmabuf IPC: RCCL's peer mappings between ranks need it on this image.  HSA reads the variable when the runtime is
# initialised (lazily, at the first HIP call), so setting it here - before torch is imported, in EVERY process: a rank
# started by the driver's own torch.distributed.run as much as one started by launch_ranks() - is early enough.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

CLOCK_WARMUP_LAUNCHES = 3000      # ~20 ms of the step kernel on a scratch handle before the warmup steps (see time_env_steps)
WATCHDOG_EXIT_CODE = 3           # exit status of a run whose secondary leg wedged (see main)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def algorithmic_bytes_per_env_step(n: int, rec_bytes: int) -> int:
    """DESIGN.md 'algorithmic bytes': record read+written once, actions read, outputs written.
    record = em_record_bytes (432 B at n=10); inputs placement i32 + resource f32 = 8n;
    outputs dense obs 4n(n+4) + active u64 8 + rank i8 n + reward f64 8 + reward_seen f64 8 + emitted 1 + done 1."""
    return 2 * rec_bytes + 8 * n + 4 * n * (n + 4) + 8 + n + 8 + 8 + 1 + 1


def cpu_baseline(seconds_target: float = 12.0) -> dict:
    """Times the line-for-line Python/numpy restatement of EdgeSimGym.step (oracle/) on ONE host core:
    whole 1000-step episodes with random actions, exploration disabled, until ~seconds_target."""
    from oracle import edge_env_oracle as eo
    cfg = eo.default_config()
    cfg["disable_env_exploration"] = True
    env = eo.EdgeEnvOracle(cfg, eo.LegacyDraws(0))
    n = env.n
    rs = np.random.RandomState(1)
    steps = 0
    episodes = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds_target:
        env.reset()
        for _ in range(cfg["episode_length"]):
            env.step(rs.randint(0, n, size=n), rs.uniform(0.1, 1.0, size=n).astype(np.float32))
        steps += cfg["episode_length"]
        episodes += 1
    dt = time.perf_counter() - t0
    out = {"value": steps * n / dt, "unit": "agent-steps/s", "cores": 1, "kind": "port",
           "sample": f"{episodes} episodes x 1000 steps, 1 env, random actions, exploration off, "
                     f"Python/numpy restatement (oracle/edge_env_oracle.py), {dt:.1f} s"}
    # the plain-C restatement on the same workload shape, for scale (single core)
    try:
        from oracle import c_oracle
        B = 256
        cenv = c_oracle.COracleEnv(cfg, B, seed=1)
        cenv.set_explore(True, 0)
        pl = rs.randint(0, n, size=(B, n)).astype(np.int32)
        rr = rs.uniform(0.1, 1.0, size=(B, n)).astype(np.float32)
        cenv.next_episode()
        cenv.reset()
        t1 = time.perf_counter()
        k = 0
        while time.perf_counter() - t1 < 3.0:
            for _ in range(50):
                cenv.step(pl, rr)
            k += 50
        out["c_port_value"] = k * B * n / (time.perf_counter() - t1)
    except Exception as e:  # pragma: no cover - informational only
        out["c_port_error"] = repr(e)
    # the env-plus-MAPPO loop of the reference (per-agent torch-CPU forwards, python-list buffer, full-batch update)
    try:
        from oracle import mappo_loop_oracle
        out["mappo_loop"] = mappo_loop_oracle.time_loop(8.0)
    except Exception as e:  # pragma: no cover - informational only
        out["mappo_loop_error"] = repr(e)
    # "whole host" figure: one environment per core, P independent processes of the same restatement
    try:
        import subprocess
        try:
            P = len(os.sched_getaffinity(0))
        except AttributeError:
            P = os.cpu_count() or 1
        P = max(1, min(P, int(os.environ.get("OMP_NUM_THREADS", "16")), 16))
        env_ = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        t1 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_timing", "5.0", str(i)], cwd=ROOT, env=env_,
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(P)]
        try:
            res = [json.loads(p.communicate(timeout=120)[0].strip().splitlines()[-1]) for p in procs]
        finally:                      # no child outlives the leg, whatever happened above
            for p in procs:
                if p.poll() is None:
                    p.kill()
                p.wait()
        wall = time.perf_counter() - t1
        out["whole_host"] = {"value": sum(r["steps"] * r["n"] / r["seconds"] for r in res), "unit": "agent-steps/s",
                             "processes": P, "sample": f"{P} processes x 5.0 s, one environment each, same restatement",
                             "wall_s": wall}
    except Exception as e:  # pragma: no cover - informational only
        out["whole_host_error"] = repr(e)
    out["host"] = {"nproc": os.cpu_count(), "cpu": _cpu_model()}
    # how representative the port is: the imported reference and the port timed side by side in the build container
    # (oracle/time_vs_reference.py; the reference cannot travel to this box).  ratio > 1: the port is FASTER than the reference
    try:
        pv = json.load(open(os.path.join(ROOT, "profiles", "port_vs_reference.json")))
        out["port_vs_reference"] = {k: {"ratio": v["ratio"], "reference": v["reference"], "port": v["port"], "unit": v["unit"]}
                                    for k, v in pv.items() if isinstance(v, dict) and "ratio" in v}
        out["port_vs_reference"]["measured"] = "build container, 8 cores, oracle/time_vs_reference.py"
    except Exception as e:  # pragma: no cover - informational only
        out["port_vs_reference"] = {"error": repr(e)}
    return out


def mappo_leg(device, rank, world, n_envs, iters=2):
    """configs[2]/[3]: full MAPPO loop, n_envs environments per GPU.  Returns the `mappo` object (rank 0) or None."""
    import torch.distributed as dist
    from marl_amd.config import ENV_CONFIG
    from marl_amd.trainer import MAPPOTrainer
    tr = MAPPOTrainer(dict(ENV_CONFIG), seed=0, device=device, n_envs=n_envs, rank=rank, world_size=world, verbose=False)
    T = tr.env.episode_len

    def iteration():
        t0 = time.perf_counter()
        tr.train_episode(explore=True, record_buffer=True)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        entries, states = tr.mappo.buffer.totals()
        tr.mappo.update()
        torch.cuda.synchronize(device)
        ep = float(tr.mappo.num_sgd_iter)
        # three ways to count one epoch of the update's two fused forward+backward kernels over this rank's rows:
        #  survey  SURVEY.md 8d's convention, 3 x forward FLOPs (it charges a first-layer dX product nobody computes)
        #  useful  forward + every weight gradient + dX of layers 2 and 3: actor 32,384, critic 60,800 FLOP per row
        #  issue   matrix-core instructions actually issued (padded 16-row tiles: 272 / 528 v_mfma_f32_16x16x4_f32 per tile,
        #          32 cycles each on one of 1,024 SIMDs) as seconds at the 2.4 GHz the chip sustains
        flops["survey"] += 3.0 * ep * (11392.0 * entries + 26240.0 * states)
        flops["useful"] += ep * (32384.0 * entries + 60800.0 * states)
        flops["issue_s"] += ep * (272.0 * ((entries + 15) // 16) + 528.0 * ((states + 15) // 16)) * 32.0 / (1024.0 * 2.4e9)
        return entries, t1 - t0, time.perf_counter() - t1

    flops = {"survey": 0.0, "useful": 0.0, "issue_s": 0.0}
    iteration()                                   # warmup: allocator, code objects, RCCL channels
    flops = {"survey": 0.0, "useful": 0.0, "issue_s": 0.0}
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    entries = 0
    roll = upd = 0.0
    for _ in range(iters):
        e, r, u = iteration()
        entries += e
        roll += r
        upd += u
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    # per-kernel share of a rollout step, outside the timed region: one more episode with three events per step
    tr.profile_events = []
    t_ep = time.perf_counter()
    tr.train_episode(explore=True, record_buffer=True)
    torch.cuda.synchronize(device)
    t_ep = time.perf_counter() - t_ep
    evs, tr.profile_events = tr.profile_events, None
    pol_us = sum(a.elapsed_time(b) for a, b, _ in evs) * 1e3 / len(evs)
    env_us = sum(b.elapsed_time(c) for _, b, c in evs) * 1e3 / len(evs)
    e_ep, s_ep = tr.mappo.buffer.totals()
    tr.mappo.buffer.clear()
    env_bytes = algorithmic_bytes_per_env_step(tr.n_agents, tr.env.record_bytes) * n_envs
    rollout_roofline = {
        "per_step_gpu_us": {"env_step": env_us, "index_and_actor": pol_us, "sum": env_us + pol_us},
        "env_step": {"bound": "hbm", "achieved": env_bytes / (env_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": env_bytes / (env_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "bytes_per_launch": env_bytes},
        "index_and_actor": {"bound": "mfma", "achieved": 11392.0 * e_ep / T / (pol_us * 1e-6) / 1e12, "peak": 157.3,
                            "unit": "TFLOP/s", "frac": 11392.0 * e_ep / T / (pol_us * 1e-6) / 1e12 / 157.3,
                            "rows_per_launch": e_ep / T,
                            "note": "actor forward FLOPs (11,392 per acting agent) over the whole launch, which also numbers and "
                                    "copies the step's entries and samples/formats the actions: latency bound, not MFMA bound"},
        "episode_wall_s_with_events": t_ep,
        "note": "HIP events on the launch stream around each launch of one extra episode (3 events per step); the two "
                "launches of a step are dependent and run one after the other"}
    stat = torch.tensor([elapsed, float(entries)], device=device, dtype=torch.float64)
    if world > 1:
        tmax = stat.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stat, op=dist.ReduceOp.SUM)
        elapsed, entries = float(tmax[0].item()), float(stat[1].item())
    tr.env.close()
    if rank != 0:
        return None
    return {"metric": "MAPPO samples/sec (transitions through rollout + GAE + 4-epoch PPO update)",
            "value": entries / elapsed, "unit": "transitions/s", "env_steps_per_s": world * n_envs * T * iters / elapsed,
            "agent_steps_per_s": world * n_envs * T * tr.n_agents * iters / elapsed, "n_envs_per_gpu": n_envs,
            "iterations": iters, "s_per_iteration": elapsed / iters, "rollout_s_rank0": roll / iters,
            "update_s_rank0": upd / iters, "dtype": "f32 (MFMA) / f64 env", "parallelism": f"dp{world}",
            # the update is the GEMM-shaped part: algorithmic FLOPs of rank 0's actor/critic forward + backward over its
            # wall time (GAE, reductions and Adam included) against the dense f32 matrix-core peak
            "update_roofline": {"bound": "mfma", "achieved": flops["useful"] / max(upd, 1e-9) / 1e12, "peak": 157.3,
                                "unit": "TFLOP/s", "frac": flops["useful"] / max(upd, 1e-9) / 1e12 / 157.3,
                                "mfma_issue_frac": flops["issue_s"] / max(upd, 1e-9),
                                "survey_convention_frac": flops["survey"] / max(upd, 1e-9) / 1e12 / 157.3,
                                "note": "achieved/frac: useful FLOPs (forward + weight gradients + dX of layers 2-3; actor "
                                        "32,384, critic 60,800 per row and epoch) over the update's whole wall time (GAE, "
                                        "reductions, Adam included); mfma_issue_frac: time the issued MFMA instructions need "
                                        "at 2.4 GHz over the same wall time; survey_convention_frac: 3 x forward FLOPs"},
            "rollout_roofline": rollout_roofline,
            "rollout": "two launches per step (env step; entry numbering + copy + actor in one kernel), states by reference, "
                       "acting-agent totals from the step kernel, one critic launch per episode (inside rollout_s)",
            "update": "4 epochs x (actor + critic fused forward/loss/backward); the critic kernel runs two wavefronts per SIMD "
                      "(dW1 and the tile staging in the second one) and its one-output head on the vector ALU; PMC MFMA busy "
                      "0.73 critic / 0.76 actor, profiles/r03_mlp_bwd_pmc_sq.txt)",
            "collective": "1 fused all-reduce of 19,103 floats per epoch + 3 doubles per update" if world > 1 else "none"}


def ippo_leg(device, rank, world, n_envs=16384, iters=2, algorithm="ippo"):
    """configs[4]: independent PPO on the scaled topology (64 servers x 256 deciders, 10x arrival rate), n_envs
    environments per GPU; one iteration = 1000-step rollout (256 per-agent actors and critics) + per-agent GAE + 4 PPO
    epochs; data parallel over the ranks (one all-reduce of the [256, 21.5k] gradient buffer per epoch)."""
    import torch.distributed as dist
    from marl_amd.ippo import IPPOTrainer
    tr = IPPOTrainer(None, seed=0, device=device, n_envs=n_envs, rank=rank, world_size=world, verbose=False, algorithm=algorithm)
    T, D = tr.env.episode_len, tr.D

    def iteration():
        t0 = time.perf_counter()
        tr.train_episode(explore=True, record_buffer=True)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        e_local = tr.algo.buffer.total() if tr.algo.buffer.steps_recorded else 0
        out = tr.algo.update()
        torch.cuda.synchronize(device)
        work["entries"] += e_local
        work["epochs"] = float(tr.algo.num_sgd_iter)
        return out.get("transitions", 0), t1 - t0, time.perf_counter() - t1

    work = {"entries": 0, "epochs": 4.0}
    iteration()
    work["entries"] = 0
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    entries, roll, upd = 0, 0.0, 0.0
    for _ in range(iters):
        e, r, u = iteration()
        entries += e            # already global (moments are all-reduced)
        roll += r
        upd += u
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    tr.env.close()
    del tr
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    # useful FLOPs per row and epoch (forward + every weight gradient + dX of layers 2 and 3), rank 0's rows over its update
    # wall time (GAE, grouping, reductions, Adam and the collectives included) against the dense f32 matrix-core peak.
    # grouped actor 68-64-64-65: 2 x 25,216 + 16,512; local critic 68-64-64-1: 2 x 17,024 + 8,320; central critic: sparse
    # first layer 2 x 8,704 per entry + head (layers 2-3) 24,960 per (step, env) state - counted per entry here, an upper
    # bound of 1 state per entry
    per_row = (66944.0 + 42368.0) if algorithm != "mappo" else (66944.0 + 17408.0 + 24960.0)
    useful = work["epochs"] * per_row * work["entries"]
    update_roofline = {"bound": "mfma", "achieved": useful / max(upd, 1e-9) / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                       "frac": useful / max(upd, 1e-9) / 1e12 / 157.3, "flop_per_row_epoch": per_row,
                       "note": "useful FLOPs (forward + weight gradients + dX of layers 2-3) of rank 0's rows over the update's "
                               "whole wall time; padding of 68 -> 80 inputs and 65 -> 80 outputs is not counted as work"}
    if algorithm == "mappo":
        return {"update_roofline": update_roofline, "metric": "MAPPO samples/sec (scaled topology: shared actor + central 17,408-wide critic, sparse first layer)",
                "value": entries / elapsed, "unit": "transitions/s", "env_steps_per_s": world * n_envs * T * iters / elapsed,
                "agent_steps_per_s": world * n_envs * T * 256 * iters / elapsed, "n_envs_per_gpu": n_envs, "iterations": iters,
                "s_per_iteration": elapsed / iters, "rollout_s_rank0": roll / iters, "update_s_rank0": upd / iters,
                "dtype": "f32 (MFMA) / f64 env", "parallelism": f"dp{world}",
                "parity": "topology is the build's generalisation; arithmetic pinned to the dense torch restatement",
                "collective": "1 all-reduce of 1,131,273 floats per epoch + 3 doubles per update" if world > 1 else "none"}
    return {"update_roofline": update_roofline,
            "metric": "IPPO samples/sec (scaled topology: transitions through rollout + per-agent GAE + 4-epoch PPO update)",
            "value": entries / elapsed, "unit": "transitions/s", "env_steps_per_s": world * n_envs * T * iters / elapsed,
            "agent_steps_per_s": world * n_envs * T * D * iters / elapsed, "n_envs_per_gpu": n_envs, "learners": D,
            "iterations": iters, "s_per_iteration": elapsed / iters, "rollout_s_rank0": roll / iters,
            "update_s_rank0": upd / iters, "dtype": "f32 (MFMA) / f64 env", "parallelism": f"dp{world}",
            "parity": "unpinned vs the reference (RLlib IPPO not vendored); pinned to oracle/ippo_oracle.py",
            "collective": "1 all-reduce of [256, 21,513] floats per epoch + [256, 3] doubles per update" if world > 1 else "none"}


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_env_steps(device, rank, world, B, steps, warmup):
    """Times `steps` launches of em_step over B environments per GPU (configs[1] protocol): warmup, barrier +
    synchronize, HIP events on the launch stream around the timed region, barrier + synchronize, MAX over ranks.
    Returns (wall seconds, device milliseconds between the events, n_nodes, episode length, record bytes, kernel variant)."""
    import torch.distributed as dist
    from marl_amd.config import ENV_CONFIG
    from marl_amd.env import BatchedEdgeSimGym

    cfg = dict(ENV_CONFIG)
    cfg["disable_env_exploration"] = True
    env = BatchedEdgeSimGym(cfg, n_envs=B, device=device, seed=0, env_offset=rank * B)
    n = env.n_nodes
    T = env.episode_len
    # pre-generated random actions, resident in HBM: placement ~ U{0..n-1}, resource ~ U(0.1, 1.0)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    POOL = 16
    pool = [(torch.randint(0, n, (B, n), device=device, generator=gen, dtype=torch.int32),
             (torch.rand((B, n), device=device, generator=gen) * 0.9 + 0.1).contiguous()) for _ in range(POOL)]
    ptrs = [(p.data_ptr(), r.data_ptr()) for p, r in pool]
    L, h, out = env._L, env._h, C.byref(env._out)
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream

    # argument tuples converted once: in a 20-launch burst the host's 7-8 us per call, not the GPU's 7, paces the region
    step_fn = L.em_step
    call_args = [(h, C.c_void_p(pp), C.c_void_p(rr), None, out, C.c_void_p(sptr)) for pp, rr in ptrs]

    def run(k, counter):
        c = counter[0]
        for _ in range(k):
            if c % T == 0:
                env.reset()
            if step_fn(*call_args[c % POOL]):
                raise RuntimeError(L.em_last_error().decode())
            c += 1
        counter[0] = c

    # Clock warm-up: the driver's default is 5 warmup + 20 timed steps, 175 us in all, which the GPU spends ramping up from its
    # idle clock.  A scratch handle (its own state, same kernel, same batch) runs CLOCK_WARMUP_LAUNCHES steps first; the W warmup
    # steps and the K timed steps of the measured environment follow at once.  Reported under config.clock_warmup.
    scratch_env = BatchedEdgeSimGym(cfg, n_envs=B, device=device, seed=12345, env_offset=rank * B)
    scratch_env.reset()
    sh, sout = scratch_env._h, C.byref(scratch_env._out)
    counter = [0]
    for i in range(CLOCK_WARMUP_LAUNCHES):
        pp, rr = ptrs[i % POOL]
        if (i + 1) % T == 0:
            scratch_env.reset()
        L.em_step(sh, pp, rr, None, sout, sptr)
    # the W warmup steps go through the same bracket as the timed ones (events, synchronize), so that the host side of the
    # timed region (event objects, method caches) is warm as well
    w0, w1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    w0.record(stream)
    run(warmup, counter)
    w1.record(stream)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the synchronize above parked this thread; a few milliseconds of spinning bring its core back to speed before the clock
    # starts (without it the first twenty launch calls took 75 us in some runs and 140-180 us in others)
    t_spin = time.perf_counter() + 0.005
    while time.perf_counter() < t_spin:
        pass
    t0 = time.perf_counter()
    ev0.record(stream)
    run(steps, counter)
    t_sub = time.perf_counter()
    ev1.record(stream)
    # Each rank's clock stops when ITS work is done; the barrier that follows orders the ranks, and the MAX over ranks
    # below is the job's time (the driver times as few as 20 steps, 140 us of GPU work: a barrier inside the clock would
    # add one collective to that).  Polling the closing event instead of synchronize() was measured and is slower (8.8 vs
    # 7.8 us per step over 20 launches).
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    dev_ms = ev0.elapsed_time(ev1)
    if rank == 0:      # diagnostic: how long the host took to queue the timed launches (the GPU starves if this exceeds its own time)
        print(f"[bench] {B} envs: host queued {steps} launches in {(t_sub - t0) * 1e6:.1f} us, GPU {dev_ms * 1e3:.1f} us between the events, "
              f"wall {elapsed * 1e6:.1f} us", file=sys.stderr, flush=True)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    # parity spot check outside the timed region: a finite reward
    assert bool(torch.isfinite(env.reward).all())
    record_bytes = env.record_bytes
    variant = env.kernel_variant()
    env.close()
    scratch_env.close()
    del pool
    return elapsed, dev_ms, n, T, record_bytes, variant


def fused_leg(device, rank, world, B, k_steps=250, launches=8):
    """em_step_many: k_steps consecutive steps per launch on pre-generated actions (open-loop callers: scripted or
    random actions, the greedy baseline).  Same workload as the headline otherwise; reported beside it, not as `value`."""
    import torch.distributed as dist
    from marl_amd.config import ENV_CONFIG
    from marl_amd.env import BatchedEdgeSimGym
    cfg = dict(ENV_CONFIG)
    cfg["disable_env_exploration"] = True
    env = BatchedEdgeSimGym(cfg, n_envs=B, device=device, seed=0, env_offset=rank * B)
    n, T = env.n_nodes, env.episode_len
    gen = torch.Generator(device=device).manual_seed(4321 + rank)
    pl = torch.randint(0, n, (k_steps, B, n), device=device, generator=gen, dtype=torch.int32)
    rs = (torch.rand((k_steps, B, n), device=device, generator=gen) * 0.9 + 0.1).contiguous()
    per_episode = T // k_steps

    def run(count):
        for i in range(count):
            if i % per_episode == 0:
                env.reset()
            env.step_many(k_steps, placement=pl, resource=rs)

    run(per_episode)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(launches)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.close()
    steps = launches * k_steps
    return {"entry_point": "em_step_many", "steps_per_launch": k_steps, "n_envs_per_gpu": B, "steps": steps,
            "value": world * B * n * steps / elapsed, "unit": "agent-steps/s", "us_per_step": elapsed * 1e6 / steps,
            "outputs": "reward/done history [k][B] + last observation", "note": "state records stay in LDS between steps"}


def scaled_leg(device, rank, world, B=65536, steps=300):
    """configs[4] environment: 64 edge servers x 256 decider agents, 10x arrival rate (ems_step, one wavefront per
    environment, compact outputs).  Random per-slot actions resident in HBM; reported beside the headline."""
    import torch.distributed as dist
    from marl_amd.config import SCALED_ENV_CONFIG
    from marl_amd.env_scaled import PENDING_CAP, ScaledEdgeSimGym
    cfg = dict(SCALED_ENV_CONFIG)
    cfg["disable_env_exploration"] = True
    env = ScaledEdgeSimGym(cfg, n_envs=B, device=device, seed=0, env_offset=rank * B)
    S, D, T = env.n_nodes, env.n_deciders, env.episode_len
    gen = torch.Generator(device=device).manual_seed(99 + rank)
    pool = [(torch.randint(0, S, (B, PENDING_CAP), device=device, generator=gen, dtype=torch.int32),
             (torch.rand((B, PENDING_CAP), device=device, generator=gen) * 0.9 + 0.1).contiguous()) for _ in range(8)]
    ptrs = [(p.data_ptr(), r.data_ptr()) for p, r in pool]
    L, h, out = env._L, env._h, C.byref(env._out)
    stream = torch.cuda.current_stream(device)
    sptr = stream.cuda_stream
    ksum = torch.zeros((), dtype=torch.int64, device=device)

    def run(k, counter, count_pending=False):
        for _ in range(k):
            if counter[0] % T == 0:
                env.reset()
            pp, rr = ptrs[counter[0] % 8]
            if L.ems_step(h, pp, rr, 1, out, sptr):
                raise RuntimeError(L.em_last_error().decode())
            if count_pending:
                ksum.add_(env.n_pending.sum())
            counter[0] += 1

    counter = [0]
    run(40, counter, True)
    kbar = ksum.item() / (40.0 * B)              # mean deciders holding a task per step
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # the synchronize above parked this thread; a few milliseconds of spinning bring its core back to speed before the clock
    # starts (without it the first twenty launch calls took 75 us in some runs and 140-180 us in others)
    t_spin = time.perf_counter() + 0.005
    while time.perf_counter() < t_spin:
        pass
    t0 = time.perf_counter()
    ev0.record(stream)
    run(steps, counter)
    t_sub = time.perf_counter()
    ev1.record(stream)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.close()
    # algorithmic bytes per env-step: fixed record part read+written, pending entries read and written,
    # actions of the acting slots, compact outputs (observation row + decider id per pending decider, 22 B of scalars)
    per_step = 2 * (12 * S + 64) + 28 * 2 * kbar + 8 * kbar + kbar * (4 * (S + 4) + 4) + 22
    ach = per_step * B / (dev_ms * 1e-3 / steps) / 1e9
    return {"workload": "scaled topology: 64 servers x 256 deciders, 10x arrival rate, random actions, step-only",
            "entry_point": "ems_step", "n_envs_per_gpu": B, "steps": steps, "env_steps_per_s": world * B * steps / elapsed,
            "value": world * B * D * steps / elapsed, "unit": "agent-steps/s", "mean_pending_per_step": kbar,
            "avg_launch_us": dev_ms * 1e3 / steps,
            "roofline": {"bound": "hbm", "kernel": "scaled_env4_kernel<step>", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic_for(f"scaled_env4_kernel@{B}", "scaled")[0],
                         "traffic_source": traffic_for(f"scaled_env4_kernel@{B}", "scaled")[1],
                         "bytes_per_env_step": per_step, "bytes_per_launch": per_step * B}}


def marl_env_leg(device, rank, world, B=65536, steps=400, lanes=0):
    """EdgeMARLEnv (the PyMARL twin environment): emm_step, one lane per environment, discrete random
    actions resident in HBM, in-kernel Philox stream.  Reported beside the headline."""
    import torch.distributed as dist
    from marl_amd.marl_env import EdgeMARLEnv
    env = EdgeMARLEnv(n_envs=B, device=device, env_offset=rank * B, seed=0)
    env.set_lanes_per_env(lanes)
    n, T = env.n_agents, env.episode_limit
    gen = torch.Generator(device=device).manual_seed(7 + rank)
    pool = [torch.randint(0, n, (B, n), device=device, generator=gen, dtype=torch.int32) for _ in range(8)]
    stream = torch.cuda.current_stream(device)
    counter = [0]
    ptrs = [p.data_ptr() for p in pool]
    env._out.info = env._log[0].data_ptr()
    L, h, out, sptr = env._L, env._h, C.byref(env._out), stream.cuda_stream

    def run(k):
        for _ in range(k):
            if counter[0] % T == 0:
                env.reset()
                env._out.info = env._log[0].data_ptr()
            if L.emm_step(h, ptrs[counter[0] % 8], None, None, out, sptr):
                raise RuntimeError(L.em_last_error().decode())
            counter[0] += 1

    run(40)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run(steps)
    ev1.record(stream)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    env.close()
    # algorithmic bytes per env-step: state arrays read and written (load, workload, data, deadline f64 +
    # queue f32 per agent; clock f64 + step i32), actions read, observation tensor + reward + flag + 6 info scalars written
    per_step = 2 * (n * 36 + 12) + 4 * n + 4 * n * (n + 4) + 8 + 1 + 48
    ach = per_step * B / (dev_ms * 1e-3 / steps) / 1e9
    return {"workload": "EdgeMARLEnv (PyMARL twin env), edge_marl.yaml defaults, discrete random actions, step-only",
            "entry_point": "emm_step", "n_envs_per_gpu": B, "steps": steps, "env_steps_per_s": world * B * steps / elapsed,
            "value": world * B * n * steps / elapsed, "unit": "agent-steps/s", "avg_launch_us": dev_ms * 1e3 / steps,
            "roofline": {"bound": "hbm", "kernel": "marl_env_kernel<step>", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic_for(f"marl_env_kernel@{B}", "marl")[0],
                         "traffic_source": traffic_for(f"marl_env_kernel@{B}", "marl")[1],
                         "bytes_per_env_step": per_step, "bytes_per_launch": per_step * B}}


# the three gradient messages of the data-parallel updates (floats per all-reduce): default-topology MAPPO (19,095
# parameters + 8 loss-sum slots), scaled-topology MAPPO (shared actor + 17,408-wide central critic + loss sums) and IPPO
# (256 learners x (actor 12,866 + 3 + critic 8,641 + 3)); tests/test_bench_cpu.py holds them to the trainers' buffers
def _mlp_params(i, o, log_std=False):
    return i * 64 + 64 + 64 * 64 + 64 + 64 * o + o + (o if log_std else 0)


COLLECTIVE_MESSAGES = (
    ("mappo_grads", (_mlp_params(14, 11, True) + _mlp_params(140, 1) + 8,)),                        # 19,103
    ("mappo_scaled_grads", (1, _mlp_params(68, 65, True) + 3 + _mlp_params(256 * 68, 1) + 3)),      # 1,131,273
    ("ippo_grads", (256, _mlp_params(68, 65, True) + 3 + _mlp_params(68, 1) + 3)))                  # [256, 21,513]


def collective_probe(device, world, reps=10, warm=3, messages=None):
    """The `collective` object of an N-rank line: which backend carried the run, how many ranks it saw (a SUM of ones
    must come back as the world size on every rank) and the duration of each of the three real gradient all-reduces in
    isolation, from HIP events on the launch stream (torch orders the current stream behind the collective) with the host
    clock beside it.  Every rank runs it; MAX over ranks is reported.  (A CPU device - the gloo test rig - has no events:
    host clock only.)"""
    import torch.distributed as dist
    device = torch.device(device)
    gpu = device.type == "cuda"
    backend = dist.get_backend()
    drain = gpu and backend != "nccl"   # gloo stages through the host from its own threads (see marl_amd/parallel.py)
    stream = torch.cuda.current_stream(device) if gpu else None

    def sync():
        if gpu:
            torch.cuda.synchronize(device)

    ones = torch.ones(1, dtype=torch.float32, device=device)
    sync()
    dist.all_reduce(ones)
    seen = int(round(float(ones.item())))
    out = {"backend": backend, "is_rccl": backend == "nccl", "world_size": dist.get_world_size(), "ranks_seen_by_sum_of_ones": seen,
           "placement": "all ranks on one GPU (rehearsal)" if os.environ.get("EDGEMARL_BENCH_DEVICE") is not None else "one GPU per rank",
           "messages": {}}
    try:
        out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:  # pragma: no cover - informational only
        out["rccl_version"] = None
    for name, shape in (COLLECTIVE_MESSAGES if messages is None else messages):
        buf = torch.ones(shape, dtype=torch.float32, device=device)
        for _ in range(warm):
            if drain:
                stream.synchronize()
            dist.all_reduce(buf)
            buf.fill_(1.0)
        sync()
        dist.barrier()
        sync()
        if gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        t0 = time.perf_counter()
        for _ in range(reps):
            if drain:
                stream.synchronize()
            dist.all_reduce(buf)
        if gpu:
            e1.record(stream)
        sync()
        host = time.perf_counter() - t0
        ev = e0.elapsed_time(e1) * 1e3 / reps if gpu else float("nan")
        tt = torch.tensor([ev if gpu else 0.0, host * 1e6 / reps], dtype=torch.float64, device=device)
        if drain:
            stream.synchronize()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        nbytes = buf.numel() * 4
        ev_us, host_us = (float(tt[0].item()) if gpu else None), float(tt[1].item())
        # bus bandwidth of an all-reduce in RCCL's own convention: 2 (N - 1) / N x bytes / time
        out["messages"][name] = {"floats": buf.numel(), "bytes": nbytes, "event_us": ev_us, "host_us": host_us,
                                 "bus_GBs": 2.0 * (world - 1) / world * nbytes / max(ev_us if gpu else host_us, 1e-3) / 1e3}
        del buf
    return out


class LineGuard:
    """The JSON line and its watchdog.  The headline exists before the secondary legs run; they only add objects to it.  A
    leg that never returns (a wedged collective or GPU sync cannot be caught as an exception) is a FAILED run: after
    `limit` seconds the watchdog prints what there is, for the record, and ends the process with WATCHDOG_EXIT_CODE.
    `line` is shared between the main thread and the watchdog thread: every mutation and the serialisation go through one
    lock, the watchdog serialises a snapshot, and its body ends in os._exit whatever happens on the way."""

    def __init__(self, line: dict, rank: int, limit_s: float):
        self.line, self.rank, self.limit = line, int(rank), float(limit_s)
        self.lock, self.printed = threading.Lock(), threading.Lock()
        self.current_leg = ["start"]
        self.timer = threading.Timer(self.limit, self._watchdog)
        self.timer.daemon = True

    def start(self):
        self.timer.start()

    def put(self, key, value):
        with self.lock:
            self.line[key] = value

    def emit(self, note=None):
        if not self.printed.acquire(blocking=False):
            return
        with self.lock:
            if note:
                self.line["legs_cut_short"] = note
            text = json.dumps(dict(self.line))
        if self.rank == 0:
            print(text, flush=True)

    def _watchdog(self):
        try:
            self.emit(f"watchdog after {self.limit:.0f} s: leg '{self.current_leg[0]}' did not return; objects present so far "
                      f"are complete; exit status {WATCHDOG_EXIT_CODE}")
            print(f"[bench] rank {self.rank}: watchdog after {self.limit:.0f} s in leg '{self.current_leg[0]}'",
                  file=sys.stderr, flush=True)
            time.sleep(2.0 if self.rank == 0 else 6.0)      # rank 0's line first, then everyone leaves
        finally:
            os._exit(WATCHDOG_EXIT_CODE)

    def finish(self):
        self.current_leg[0] = "done"
        self.timer.cancel()
        if self.timer.ident is not None:  # started: a cancelled Timer thread ends at once, nothing outlives main()
            self.timer.join(timeout=1.0)
        self.emit()


def guarded(name, fn, *a, **k):
    """Secondary legs (everything beside the headline measurement) must not cost the headline line: a Python-level
    failure is reported in the leg's object and on stderr.  The same code runs on every rank, so a deterministic failure
    is caught on all of them and no rank is left waiting in a collective."""
    try:
        return fn(*a, **k)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] {name} leg failed: {e!r}", file=sys.stderr, flush=True)
        return {"error": repr(e)[:300]}


_SRC = {"env": ("edge_env.hip", "em_env_math.h", "em_common.h"),
        "scaled": ("edge_env_scaled.hip", "em_env_math.h", "em_common.h"),
        "marl": ("edge_marl_env.hip", "em_env_math.h", "em_common.h")}


def kernel_signature(which: str = "env") -> str:
    """Hash of a step kernel's sources: a PMC traffic figure is only echoed for the code it was measured on."""
    import hashlib
    hsh = hashlib.sha1()
    for name in _SRC[which]:
        with open(os.path.join(ROOT, "marl_amd", "csrc", name), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()[:16]


def traffic_for(key, which: str = "env"):
    """HBM bytes per launch of a step kernel from the separate rocprofv3 --pmc passes of tools/measure_traffic.sh (the
    counters cannot be read from inside this process).  profiles/traffic.json records, per entry, the hash of the kernel
    sources it was measured on; a figure taken on other kernel code is omitted (null), never echoed.  Returns
    (bytes or None, provenance string or None)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        doc = json.load(open(tpath))
        key = str(key)
        if doc.get("signatures", {}).get(key) != kernel_signature(which):
            return None, None
        v = doc.get(key)
        return (v, f"profiles/traffic.json ({doc.get('measured', 'tools/measure_traffic.sh')})") if v is not None else (None, None)
    except Exception:
        return None, None


def launch_command(n_gpus: int, argv: list, port: int) -> list:
    """Command line of the N-rank job that `python bench.py --gpus N` starts when it is not already a rank: the same
    launcher the driver uses (torch.distributed.run, one process per GPU, rendezvous on 127.0.0.1), the same argv."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n_gpus)}",
            "--master-addr", "127.0.0.1", "--master-port", str(int(port)), os.path.abspath(__file__)] + list(argv)


def free_port() -> int:
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def launch_ranks(n_gpus: int, argv: list) -> int:
    """Parent of an N-rank run.  It never touches the GPU (no HIP call, no torch.cuda query before this point): the
    ranks are CHILD processes and the parent only waits for them and passes their exit code on."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's peer mappings need it on this image
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = launch_command(n_gpus, argv, free_port())
    print("[bench] launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mappo", action="store_true", help="skip the full-loop MAPPO measurement")
    ap.add_argument("--no-scale-point", action="store_true", help="skip the extra 65,536-environment measurement")
    ap.add_argument("--mappo-envs", type=int, default=65536, help="environments per GPU of the MAPPO measurement")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s); running with {world}", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rehearsal of the N > 1 code path on a one-GPU box: EDGEMARL_BENCH_BACKEND=gloo EDGEMARL_BENCH_DEVICE=0 puts every
        # rank on one card over gloo (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
        backend = os.environ.get("EDGEMARL_BENCH_BACKEND", "nccl")
        local_rank = int(os.environ.get("EDGEMARL_BENCH_DEVICE", local_rank))
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    device = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(device)

    B = args.envs
    elapsed, dev_ms, n, T, record_bytes, variant = time_env_steps(device, rank, world, B, args.steps, args.warmup)

    # the headline line exists from here on; the legs below only add objects to it.  A watchdog prints what there is and
    # ends the process if a secondary leg wedges (a collective that never returns cannot be caught as an exception), so a
    # hang beside the headline costs the legs, not the line.
    agent_steps = world * B * n * args.steps
    bytes_launch = algorithmic_bytes_per_env_step(n, record_bytes) * B
    launch_s = dev_ms * 1e-3 / args.steps          # HIP events on the launch stream, reset launches included
    achieved = bytes_launch / launch_s / 1e9
    traffic, traffic_src = traffic_for(B)
    line = {
        "metric": "agent-steps/sec (env step)",
        "value": agent_steps / elapsed,
        "unit": "agent-steps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": f"EdgeSimGym default topology, {B} batched envs/GPU, random actions, step-only kernel",
                   "n_envs_per_gpu": B, "n_nodes": n, "episode_length": T, "rng": "philox4x32-10 in-kernel",
                   "clock_warmup": f"{CLOCK_WARMUP_LAUNCHES} launches of the same kernel on a scratch handle before the warmup steps",
                   "exploration": "disabled", "lanes_per_env": variant["lanes_per_env"],
                   "workgroup": f"{variant['workgroup_waves'] - 1} worker + 1 scalar wavefronts",
                   "parallelism": f"dp{world}" if world > 1 else "single GPU"},
        "roofline": {"bound": "hbm", "kernel": variant["kernel"], "achieved": achieved,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": traffic_src, "bytes_per_launch": bytes_launch, "avg_launch_us": launch_s * 1e6},
    }
    guard = LineGuard(line, rank, float(os.environ.get("EDGEMARL_BENCH_WATCHDOG_S", "900")))
    put, emit, current_leg = guard.put, guard.emit, guard.current_leg
    guard.start()

    def leg(key, fn, *a, **k):
        current_leg[0] = key
        t_leg = time.perf_counter()
        put(key, guarded(key, fn, *a, **k))
        if rank == 0:      # progress on stderr: the line itself is printed once, at the end
            print(f"[bench] {key}: {time.perf_counter() - t_leg:.1f} s", file=sys.stderr, flush=True)

    if world > 1:
        # proof that the run's collectives went through the backend named, with N ranks, and what the three real
        # gradient messages cost on it
        leg("collective", collective_probe, device, world)
    if not args.no_scale_point and B != 65536:
        # north_star's ">= 65k concurrent envs": same kernel, same timing protocol, 16x the batch (not part of `value`)
        k = max(200, args.steps // 4)
        current_leg[0] = "scale_point"
        if rank == 0:
            print(f"[bench] headline measured ({agent_steps / elapsed:.3e} agent-steps/s); secondary legs follow", file=sys.stderr, flush=True)
        res = guarded("scale_point", time_env_steps, device, rank, world, 65536, k, max(20, args.warmup // 4))
        if isinstance(res, tuple):
            se, sd, sv = res[0], res[1], res[5]
            s_bytes = algorithmic_bytes_per_env_step(n, record_bytes) * 65536
            s_ach = s_bytes / (sd * 1e-3 / k) / 1e9
            s_traffic, s_src = traffic_for(65536)
            put("scale_point", {"n_envs_per_gpu": 65536, "steps": k, "value": world * 65536 * n * k / se,
                                "unit": "agent-steps/s", "avg_launch_us": sd * 1e3 / k,
                                "lanes_per_env": sv["lanes_per_env"], "kernel": sv["kernel"],
                                "roofline": {"bound": "hbm", "achieved": s_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": s_ach / HBM_PEAK_GBS, "traffic": s_traffic,
                                             "traffic_source": s_src, "bytes_per_launch": s_bytes}})
        else:
            put("scale_point", res)
    if not args.no_scale_point:
        leg("fused_steps", fused_leg, device, rank, world, B)
        leg("scaled_topology", scaled_leg, device, rank, world)
        leg("pymarl_env", marl_env_leg, device, rank, world)
    if not args.no_mappo:
        leg("mappo", mappo_leg, device, rank, world, args.mappo_envs)
        leg("ippo_scaled", ippo_leg, device, rank, world)
        leg("mappo_scaled", ippo_leg, device, rank, world, algorithm="mappo")
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        leg("cpu_baseline", cpu_baseline)
    guard.finish()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
