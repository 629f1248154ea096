"""Per-kernel statistics (the table of `rocprofv3 --kernel-trace --stats`) from a rocpd result database:
    python tools/kernel_stats.py gpurun_out/prof/bench_results.db > profiles/rNN_bench_full_kernel_stats.csv"""
import sqlite3
import statistics
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, start, end from kernels").fetchall()
acc = defaultdict(list)
for k, s, e in rows:
    acc[k].append(e - s)
total = sum(sum(v) for v in acc.values())
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    sd = statistics.pstdev(v) if len(v) > 1 else 0.0
    print(f'"{k}",{len(v)},{sum(v)},{sum(v) / len(v):.6f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)},{sd:.6f}')
