"""Kernel sequence of ONE update in a rocpd kernel trace: every kernel from the last `gae_kernel` launch to the end of the
trace, with its start relative to the window, its duration and the idle gap in front of it.
    python tools/update_window.py results.db [first-kernel-substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kt = "kernels" if "kernels" in tabs else [t for t in tabs if t.startswith("kernels")][0]
cols = [r[1] for r in c.execute(f"pragma table_info({kt})")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, start, end from {kt} order by start").fetchall()
key = sys.argv[2] if len(sys.argv) > 2 else "gae_kernel"
first = max(i for i, r in enumerate(rows) if key in r[0])
rows = rows[first:]
t0 = rows[0][1]
busy = 0
last_end = t0
for k, s, e in rows:
    gap = s - last_end
    busy += e - s
    short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{1e-3 * (s - t0):12.1f} us  +{1e-3 * (e - s):10.1f} us  gap {1e-3 * gap:8.1f} us  {short}")
    last_end = max(last_end, e)
print(f"window {1e-6 * (last_end - t0):.3f} ms, kernels {1e-6 * busy:.3f} ms, idle {1e-6 * (last_end - t0 - busy):.3f} ms")
