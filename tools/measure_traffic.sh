#!/bin/bash
# HBM traffic of the env step kernel from the L2 memory-side counters (MI355X_MICROARCH.md "HBM"):
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes.  Run on the GPU box from the repo root:
#   bash tools/measure_traffic.sh ; python3 tools/traffic_json.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for B in 4096 65536; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $C -d gpurun_out/pmc_${C}_$B -o pmc -- python3 bench.py --envs $B --steps 200 --warmup 20 --no-cpu-baseline --no-mappo --no-scale-point > gpurun_out/pmc_${C}_$B.log 2>&1
  done
done
