"""Static instruction mix per kernel of a gfx950 assembly file (hipcc -S --cuda-device-only): tools/isa_mix.py file.s [name-filter]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.Lfunc_end", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name:
        continue
    ins = [l.strip().split()[0] for l in body.split("\n")
           if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.strip().split()[0].endswith(":")]
    c = Counter()
    for i in ins:
        key = "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else \
            "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
        c[key] += 1
    print(name[:90], dict(c), "total", len(ins))
    print("  ", Counter(ins).most_common(14))
