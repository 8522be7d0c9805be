#!/bin/bash
# contraction grid x kernel form, host off the critical path: does a contraction that leaves CUs to the gather pay now?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c38
STEPS=300 bash experiments/env_run.sh 1 "SAGE_DENSE_BLOCKS=256" "SAGE_DENSE_BLOCKS=192" "SAGE_DENSE_BLOCKS=128" "SAGE_DENSE_BLOCKS=96" "SAGE_DENSE_BLOCKS=64" \
  "SAGE_DENSE_PC=1 SAGE_DENSE_BLOCKS=256" "SAGE_DENSE_PC=1 SAGE_DENSE_BLOCKS=192" "SAGE_DENSE_PC=1 SAGE_DENSE_BLOCKS=128" "SAGE_DENSE_PC=1 SAGE_DENSE_BLOCKS=96" "SAGE_DENSE_PC=1 SAGE_DENSE_BLOCKS=64" \
  2>&1 | cut -c1-200 | tee gpurun_out/r03c38/grid.log
