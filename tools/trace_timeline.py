"""Timeline of a rocpd kernel trace: per kernel name the launch count and mean duration, the overlap between kernels
(time during which two or more kernels were in flight) and the mean gap between consecutive kernels of one stream/queue.
    python tools/trace_timeline.py results.db [name-substring-to-window-on]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
rows = c.execute(f"select {name}, start, end, {q} from kernels order by start").fetchall()
if len(sys.argv) > 2:
    idx = [i for i, r in enumerate(rows) if sys.argv[2] in r[0]]
    rows = rows[idx[0]:idx[-1] + 1]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for k, s, e, _ in rows:
    ev += [(s, 1), (e, -1)]
ev.sort()
busy = over = 0
depth, last = 0, ev[0][0]
for t, d in ev:
    if depth >= 1:
        busy += t - last
    if depth >= 2:
        over += t - last
    depth += d
    last = t
print(f"window {1e-6 * (t1 - t0):.3f} ms, {len(rows)} kernels; some kernel in flight {1e-6 * busy:.3f} ms, two or more {1e-6 * over:.3f} ms")
acc = defaultdict(list)
for k, s, e, _ in rows:
    acc[k[:60]].append(e - s)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {k:60s} n={len(v):6d} mean {sum(v) / len(v) / 1e3:8.2f} us  total {sum(v) / 1e6:8.3f} ms")
byq = defaultdict(list)
for k, s, e, qq in rows:
    byq[qq].append((s, e))
for qq, v in byq.items():
    gaps = [b[0] - a[1] for a, b in zip(v, v[1:])]
    if gaps:
        print(f"  queue {qq}: {len(v)} kernels, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us")
