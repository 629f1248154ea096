"""Static VALU/SALU/LDS instruction count per source line of one kernel of a gfx950 assembly file built with -g1:
   tools/isa_by_line.py file.s name-filter [min-count]"""
import re
import sys
from collections import Counter, defaultdict

s = open(sys.argv[1]).read()
flt = sys.argv[2]
minc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
files = {}
for m in re.finditer(r'^\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s, re.M):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    cur = ("?", 0)
    per = defaultdict(Counter)
    tot = Counter()
    for l in body.split("\n"):
        t = l.strip()
        if not t:
            continue
        if t.startswith(".loc"):
            f = t.split()
            cur = (files.get(int(f[1]), f[1]), int(f[2]))
            continue
        if t.startswith((".", ";", "//")) or t.split()[0].endswith(":"):
            continue
        i = t.split()[0]
        key = "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem"
        per[cur][key] += 1
        tot[key] += 1
    print(name[:100], dict(tot))
    for k in sorted(per, key=lambda k: (k[0], k[1])):
        c = per[k]
        if sum(c.values()) >= minc:
            print(f"  {k[0]}:{k[1]:5d}  valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}")
