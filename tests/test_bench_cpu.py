"""bench.py's harness logic on the CPU: the environment variable RCCL's peer mappings need is set in
every process that imports bench; a wedged secondary leg ends the run with a NON-ZERO status after printing the headline;
the JSON line is serialised from a snapshot while the main thread keeps adding objects; the `collective` object of an
N-rank line (backend, ranks seen, the three gradient messages) over gloo with two ranks."""
import json
import os
import socket
import subprocess
import sys
import threading

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_import_sets_the_ipc_variable_in_every_process():
    env = {k: v for k, v in os.environ.items() if k != "HSA_ENABLE_IPC_MODE_LEGACY"}
    env["WORLD_SIZE"] = "8"          # a rank started by the driver's own torch.distributed.run
    out = subprocess.run([sys.executable, "-c", "import os, bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "0"
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"      # an explicit choice of the caller is respected
    out = subprocess.run([sys.executable, "-c", "import os, bench; print(os.environ['HSA_ENABLE_IPC_MODE_LEGACY'])"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.stdout.strip().splitlines()[-1] == "1"


_HANG = """
import threading, bench
g = bench.LineGuard({"metric": "m", "value": 1.0}, rank=0, limit_s=0.3)
g.current_leg[0] = "mappo"
g.put("scale_point", {"value": 2.0})
g.start()
threading.Event().wait()          # a leg that never returns
"""


def test_watchdog_prints_the_line_and_exits_nonzero():
    import bench
    out = subprocess.run([sys.executable, "-c", _HANG], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == bench.WATCHDOG_EXIT_CODE != 0
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["scale_point"] == {"value": 2.0}
    assert "mappo" in line["legs_cut_short"] and "watchdog" in line["legs_cut_short"]
    assert "leg 'mappo'" in out.stderr


def test_watchdog_exits_even_if_serialisation_fails():
    code = _HANG.replace('g.put("scale_point", {"value": 2.0})', 'g.put("bad", object())')      # json.dumps raises
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    import bench
    assert out.returncode == bench.WATCHDOG_EXIT_CODE


def test_line_is_serialised_from_a_snapshot_while_legs_add_objects(capsys):
    import bench
    line = {"metric": "m"}
    g = bench.LineGuard(line, rank=0, limit_s=3600.0)
    stop = threading.Event()

    def writer():
        i = 0
        while not stop.is_set():
            g.put(f"k{i % 4096}", {"i": i})
            i += 1

    th = threading.Thread(target=writer)
    th.start()
    try:
        for _ in range(200):          # what emit() does under the lock, many times against a writing thread
            with g.lock:
                json.dumps(dict(g.line))
        g.emit("note")
        g.emit("second call prints nothing")
    finally:
        stop.set()
        th.join()
    printed = capsys.readouterr().out.strip().splitlines()
    assert len(printed) == 1 and json.loads(printed[0])["legs_cut_short"] == "note"
    g.finish()                        # timer never started: cancel + join must not raise
    assert not g.timer.is_alive()


def test_collective_message_sizes():
    import bench
    m = dict(bench.COLLECTIVE_MESSAGES)
    assert m["mappo_grads"] == (5846 + 13249 + 8,)                       # actor 5,846, critic 13,249 parameters
    assert m["ippo_grads"] == (256, 12866 + 3 + 8641 + 3)
    assert m["mappo_scaled_grads"] == (1, 12866 + 3 + (64 * 17408 + 64 + 64 * 64 + 64 + 64 + 1) + 3)


def _probe_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    res = bench.collective_probe("cpu", world, reps=2, warm=1, messages=(("small", (19103,)), ("rows", (4, 100))))
    json.dump(res, open(os.path.join(out_dir, f"probe_{rank}.json"), "w"))
    dist.destroy_process_group()


def test_collective_object_two_ranks_gloo(tmp_path):
    mp.spawn(_probe_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (json.load(open(tmp_path / f"probe_{r}.json")) for r in range(2))
    for r in (r0, r1):
        assert r["backend"] == "gloo" and r["is_rccl"] is False
        assert r["world_size"] == 2 and r["ranks_seen_by_sum_of_ones"] == 2
        assert r["messages"]["small"]["floats"] == 19103 and r["messages"]["rows"]["bytes"] == 1600
        assert r["messages"]["small"]["host_us"] > 0 and np.isfinite(r["messages"]["small"]["bus_GBs"])
    assert r0["messages"]["small"]["host_us"] == r1["messages"]["small"]["host_us"]      # MAX over ranks on both
