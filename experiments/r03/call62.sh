#!/bin/bash
# contraction grid, finer, for the concat encoder (configs 3 and 5) and the gcn control
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c62
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 2 "SAGE_DENSE_BLOCKS=176" "SAGE_DENSE_BLOCKS=192" "SAGE_DENSE_BLOCKS=208" "SAGE_DENSE_BLOCKS=224" "SAGE_DENSE_BLOCKS=240" 2>&1 | cut -c1-80 | tee gpurun_out/r03c62/c3.log
STEPS=300 BENCH_ARGS="--mode concat --config 5" bash experiments/env_run.sh 2 "SAGE_DENSE_BLOCKS=192" "SAGE_DENSE_BLOCKS=208" "SAGE_DENSE_BLOCKS=224" "SAGE_DENSE_BLOCKS=240" 2>&1 | cut -c1-80 | tee gpurun_out/r03c62/c5.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 3 "SAGE_DENSE_BLOCKS=224" "SAGE_DENSE_BLOCKS=256" 2>&1 | cut -c1-80 | tee gpurun_out/r03c62/gcn20.log
