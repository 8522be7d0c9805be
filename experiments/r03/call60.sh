#!/bin/bash
# the end of the 20-step region: does the host notice the last batch late?  (interrupt vs polling wait in the final synchronize)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c60
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 4 "SAGE_X=0" "HSA_ENABLE_INTERRUPT=0" "HIP_FORCE_DEV_KERNARG=1" 2>&1 | cut -c1-90 | tee gpurun_out/r03c60/c.log
