#!/bin/bash
# concat encoder: contraction grid 256 / 224 x depth 4 / 6, configs 3 and 5 (three interleaved runs each); gcn control
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c61
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 3 "SAGE_X=0" "SAGE_DENSE_BLOCKS=224" "SAGE_DEPTH=6" "SAGE_DENSE_BLOCKS=224 SAGE_DEPTH=6" 2>&1 | cut -c1-100 | tee gpurun_out/r03c61/c3.log
STEPS=300 BENCH_ARGS="--mode concat --config 5" bash experiments/env_run.sh 2 "SAGE_X=0" "SAGE_DENSE_BLOCKS=224" "SAGE_DENSE_BLOCKS=224 SAGE_DEPTH=6" 2>&1 | cut -c1-100 | tee gpurun_out/r03c61/c5.log
STEPS=300 bash experiments/env_run.sh 3 "SAGE_X=0" "SAGE_DENSE_BLOCKS=224" 2>&1 | cut -c1-100 | tee gpurun_out/r03c61/gcn.log
