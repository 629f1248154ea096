"""Summarise a rocprofv3 --pmc result database (rocpd sqlite) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py gpurun_out/pmc_sq/pmc_results.db [kernel-substring] [skip-first-N-dispatches]
"""
import sqlite3
import sys
from collections import defaultdict


def summarise(path, needle="", skip=0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute(f"select {name_col}, counter_name, value, dispatch_id from counters_collection").fetchall()
    acc = defaultdict(lambda: defaultdict(float))
    ndisp = defaultdict(set)
    first = {}
    for k, cn, v, d in rows:
        first[k] = min(first.get(k, d), d)
    order = defaultdict(set)
    for k, cn, v, d in rows:
        order[k].add(d)
    drop = {k: set(sorted(ds)[:skip]) for k, ds in order.items()} if skip else {}
    for k, cn, v, d in rows:
        if needle and needle not in k:
            continue
        if skip and d in drop[k]:
            continue
        acc[k][cn] += v
        ndisp[k].add(d)
    out = {}
    for k in acc:
        nd = max(1, len(ndisp[k]))
        out[k] = {"dispatches": nd, **{cn: v / nd for cn, v in sorted(acc[k].items())}}
    return out


if __name__ == "__main__":
    res = summarise(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["dispatches"]):
        print(k[:110])
        for cn, x in v.items():
            print(f"    {cn:28s} {x:,.1f}")
