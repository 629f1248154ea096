"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/measure_traffic.sh into profiles/traffic.json: HBM bytes per launch
of the step kernels.  Corrections per MI355X_MICROARCH.md (HBM section): the counters are in KiB; on gfx950 FETCH_SIZE
tallies the 128-byte requests of a 16-byte-per-lane streaming read at 64 bytes, so the read side is doubled; WRITE_SIZE is
exact for 16-byte-per-lane stores.  The file records the hash of the kernel sources it was measured on (bench.py omits a
figure whose hash no longer matches)."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def signature(names):
    h = hashlib.sha1()
    for name in names:
        with open(os.path.join(ROOT, "marl_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


KERNELS = {   # key in traffic.json -> (kernel-name needle, sources whose hash pins the figure)
    "4096": ("env_kernel<16, false, false, 0, 10>", ("edge_env.hip", "em_env_math.h", "em_common.h")),
    "65536": ("env_kernel<0, false, false, 0, 10>", ("edge_env.hip", "em_env_math.h", "em_common.h")),
    "scaled_env4_kernel@65536": ("scaled_env4_kernel<false, 64, 256>", ("edge_env_scaled.hip", "em_env_math.h", "em_common.h")),
    "marl_env_kernel@65536": ("marl_env_kernel<false>", ("edge_marl_env.hip", "em_env_math.h", "em_common.h")),
}

out, detail = {}, {}
dbs = {c: os.path.join(ROOT, "gpurun_out", f"pmc_{c}", "pmc_results.db") for c in ("FETCH_SIZE", "WRITE_SIZE")}
for key, (needle, srcs) in KERNELS.items():
    vals = {}
    for c, db in dbs.items():
        res = summarise(db, needle)
        if len(res) != 1:
            print(f"{key}: {len(res)} kernels match {needle!r} in {db}: {list(res)[:3]}", file=sys.stderr)
            vals = None
            break
        (_, v), = res.items()
        vals[c] = v[c]
        vals["dispatches_" + c] = v["dispatches"]
    if not vals:
        continue
    read_b = 2.0 * vals["FETCH_SIZE"] * 1024.0
    write_b = vals["WRITE_SIZE"] * 1024.0
    out[key] = read_b + write_b
    detail[key] = {"kernel": needle, "FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
                   "read_bytes_corrected": read_b, "write_bytes": write_b, "dispatches": vals["dispatches_FETCH_SIZE"],
                   "kernel_signature": signature(srcs)}
out["kernel_signature"] = signature(KERNELS["4096"][1])
out["signatures"] = {k: d["kernel_signature"] for k, d in detail.items()}
out["measured"] = "tools/measure_traffic.sh, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, " + time.strftime("%Y-%m-%d")
out["detail"] = detail
out["note"] = ("HBM bytes per launch; read side = 2 x FETCH_SIZE (gfx950: the counter tallies 128-byte requests at 64 bytes; "
               "calibrated for 4-, 8- and 16-byte-per-lane loads with tools/calib_fetch.hip, profiles/r03_fetch_calibration.txt), "
               "write side = WRITE_SIZE")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
