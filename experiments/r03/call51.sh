#!/bin/bash
# what would a PRE-TRANSFORMED table buy (inference at fixed weights: gather rows of Y = X . W1^T, 128 floats, instead of rows of X,
# 256 floats; layer 1 = act(mean(Y[nbrs]))?  Structurally that is the same pipeline at D0 = 128 with W1 = identity: time D0 = 128.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c51
STEPS=300 BENCH_ARGS="--dim 128" bash experiments/env_run.sh 2 "SAGE_X=0" 2>&1 | cut -c1-200 | tee gpurun_out/r03c51/d128.log
STEPS=20 BENCH_ARGS="--dim 128 --warmup 5" bash experiments/env_run.sh 2 "SAGE_X=0" 2>&1 | cut -c1-200 | tee -a gpurun_out/r03c51/d128.log
