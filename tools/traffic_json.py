"""Turns the FETCH_SIZE / WRITE_SIZE passes of tools/measure_traffic.sh into profiles/traffic.json:
HBM bytes per launch of the env step kernel.  Corrections per MI355X_MICROARCH.md (HBM section):
the counters are in KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of a 16-byte-per-lane
streaming read at 64 bytes, so the read side is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
detail = {}
for B in (4096, 65536):
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        db = os.path.join(ROOT, "gpurun_out", f"pmc_{c}_{B}", "pmc_results.db")
        res = summarise(db, "env_kernel<16, false, false>")
        (k, v), = res.items()
        vals[c] = v[c]
    read_b = 2.0 * vals["FETCH_SIZE"] * 1024.0
    write_b = vals["WRITE_SIZE"] * 1024.0
    out[str(B)] = read_b + write_b
    detail[str(B)] = {"FETCH_SIZE_KiB_raw": vals["FETCH_SIZE"], "WRITE_SIZE_KiB_raw": vals["WRITE_SIZE"],
                      "read_bytes_corrected": read_b, "write_bytes": write_b}
out["detail"] = detail
out["note"] = ("per launch of env_kernel<16,philox,step>; read side = 2 x FETCH_SIZE (gfx950 correction for 16-B/lane "
               "streaming loads; the 4-B/lane action loads are uncalibrated)")
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
