#!/bin/bash
# r03 call 13: concat encoder (config 3): role maps, depth, gather blocks per CU, layer-2 waves -- is there a better operating point?
cd ${GRAFT_REPO_ROOT:-/root/repo}
BENCH_ARGS="--mode concat" STEPS=300 bash experiments/env_run.sh 2 "SAGE_ROLES=SGDL" "SAGE_ROLES=SGDD" "SAGE_ROLES=SGDL SAGE_DEPTH=6" "SAGE_ROLES=SGDL SAGE_DEPTH=8" \
  "SAGE_G_PER_CU=4" "SAGE_G_PER_CU=8" "SAGE_T16_WAVES=16" "SAGE_DENSE_BLOCKS=320" "SAGE_G_PER_CU=4 SAGE_DEPTH=6" 2>&1 | tee gpurun_out/r03c13.log
