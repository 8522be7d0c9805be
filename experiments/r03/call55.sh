#!/bin/bash
# slice-major gather: neighbours per trip (registers per wave) x blocks per CU, host threads on
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c55
STEPS=300 bash experiments/env_run.sh 1 "SAGE_X=0" "SAGE_G_TRIP=8" "SAGE_G_TRIP=8 SAGE_G_PER_CU=8" "SAGE_G_TRIP=8 SAGE_G_PER_CU=4" "SAGE_G_PER_CU=7" "SAGE_X=0" 2>&1 | cut -c1-200 | tee gpurun_out/r03c55/g.log
