#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c56
STEPS=300 bash experiments/env_run.sh 4 "SAGE_G_PER_CU=6" "SAGE_G_PER_CU=7" "SAGE_G_PER_CU=8" 2>&1 | cut -c1-60 | tee gpurun_out/r03c56/g.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 4 "SAGE_G_PER_CU=6" "SAGE_G_PER_CU=7" 2>&1 | cut -c1-60 | tee -a gpurun_out/r03c56/g.log
