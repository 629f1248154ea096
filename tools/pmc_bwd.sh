#!/bin/bash
# Matrix-core busy share and LDS stalls of the update's backward kernels (one SQ pass: 4 of the 8 SQ slots).
#   bash tools/pmc_bwd.sh   (GPU box, repo root) -> gpurun_out/pmc_bwd_r03/, summary in gpurun_out/pmc_bwd_r03.txt
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CTR="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS"
rm -rf gpurun_out/pmc_bwd_r03 gpurun_out/pmc_gbwd_r03
timeout -s KILL 240 rocprofv3 --pmc $CTR -d gpurun_out/pmc_bwd_r03 -o pmc -- python3 tools/bench_mappo.py --envs 65536 --horizon 100 --iters 1 > gpurun_out/pmc_bwd_r03.log 2>&1
timeout -s KILL 240 rocprofv3 --pmc $CTR -d gpurun_out/pmc_gbwd_r03 -o pmc -- python3 tools/bench_ippo.py --envs 16384 --horizon 60 --iters 1 > gpurun_out/pmc_gbwd_r03.log 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_bwd_r03/pmc_results.db bwd_kernel > gpurun_out/pmc_bwd_r03.txt
python3 tools/pmc_summary.py gpurun_out/pmc_gbwd_r03/pmc_results.db bwd_kernel >> gpurun_out/pmc_bwd_r03.txt
