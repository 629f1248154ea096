#!/bin/bash
# Vector / matrix / wait counters of the scaled rollout's kernels (one SQ pass of 8 counters).
#   bash tools/pmc_scaled_rollout.sh   (GPU box, repo root) -> gpurun_out/pmc_scaled_rollout_r03.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES"
rm -rf /tmp/pmc_sr
timeout -s KILL 240 rocprofv3 --pmc $CTR -d /tmp/pmc_sr -o pmc -- python3 tools/bench_ippo.py --envs 16384 --horizon 100 --iters 1 > gpurun_out/pmc_scaled_rollout_r03.log 2>&1
DB=$(ls /tmp/pmc_sr/*.db /tmp/pmc_sr/*/*.db 2>/dev/null | head -1)
for k in gmlp_fwd_kernel scaled_index scaled_env4_kernel; do python3 tools/pmc_summary.py "$DB" $k; done > gpurun_out/pmc_scaled_rollout_r03.txt
