#!/bin/bash
# Instruction mix of the environment step kernels (one SQ pass, 8 counters).
#   bash tools/pmc_env.sh   (GPU box, repo root) -> summary in gpurun_out/pmc_env_r02.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CTR="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
rm -rf gpurun_out/pmc_env_r02
: > gpurun_out/pmc_env_r02.txt
for B in 4096 65536; do
  timeout -s KILL 120 rocprofv3 --pmc $CTR -d gpurun_out/pmc_env_r02/$B -o pmc -- python3 tools/microbench_env.py $B 0 > gpurun_out/pmc_env_r02_$B.log 2>&1
  echo "== $B environments, exploration off" >> gpurun_out/pmc_env_r02.txt
  python3 tools/pmc_summary.py gpurun_out/pmc_env_r02/$B/pmc_results.db env_kernel >> gpurun_out/pmc_env_r02.txt
  rm -rf gpurun_out/pmc_env_r02/$B
done
