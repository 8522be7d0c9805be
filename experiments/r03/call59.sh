#!/bin/bash
# concat encoder (its pipeline's pacemaker is the contraction, 75 of 87 us in the pipeline): stream priorities, contraction grid, depth
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c59
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 2 "SAGE_X=0" "SAGE_PIPE_PRIO=D" "SAGE_PIPE_PRIO=DL" "SAGE_DENSE_BLOCKS=224 SAGE_DEPTH=6" "SAGE_G_PER_CU=4" "SAGE_G_PER_CU=5 SAGE_DEPTH=6" 2>&1 | cut -c1-130 | tee gpurun_out/r03c59/c.log
