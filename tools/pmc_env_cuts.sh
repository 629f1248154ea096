#!/bin/bash
# Cumulative instruction counts of the env step kernel by program point: every worker wavefront ends at stamp k.
#   bash tools/pmc_env_cuts.sh [n_envs] [lanes] [waves]   (GPU box; tools/_stamp_env built with -DEM_STAMP)
set -e
B=${1:-65536}; LANES=${2:-1}; WAVES=${3:-8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CTR="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY"
OUT=gpurun_out/pmc_env_cuts_${B}_${LANES}.txt
: > $OUT
for CUT in 1 2 3 4 5 6 9 10 11 0; do
  rm -rf gpurun_out/pmc_cut
  EM_CUT=$CUT timeout -s KILL 60 rocprofv3 --pmc $CTR -d gpurun_out/pmc_cut -o pmc -- tools/_stamp_env $B 0 $WAVES $LANES > gpurun_out/pmc_cut.log 2>&1
  echo "== cut $CUT" >> $OUT
  python3 tools/pmc_summary.py gpurun_out/pmc_cut/pmc_results.db "env_kernel<" 60 | grep -v "^void.*true" >> $OUT
done
rm -rf gpurun_out/pmc_cut
cat $OUT
