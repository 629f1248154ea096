#!/bin/bash
# HBM traffic of the step kernels from the L2 memory-side counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes over the benchmark's step-only legs (headline 4,096 envs, 65,536-env
# scale point, scaled topology, EdgeMARLEnv).  Run on the GPU box from the repo root, then build profiles/traffic.json:
#   bash tools/measure_traffic.sh && python3 tools/traffic_json.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  timeout -s KILL 300 rocprofv3 --pmc $C -d gpurun_out/pmc_$C -o pmc -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-mappo > gpurun_out/pmc_$C.log 2>&1
done
