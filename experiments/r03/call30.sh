#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS=300 bash experiments/env_run.sh 4 "SAGE_SO_THREADS=1024" "SAGE_SO_THREADS=512" "SAGE_SO_THREADS=256" 2>&1 | cut -c1-60 | tee gpurun_out/r03c30.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 4 "SAGE_SO_THREADS=1024" "SAGE_SO_THREADS=512" "SAGE_SO_THREADS=256" 2>&1 | cut -c1-60 | tee -a gpurun_out/r03c30.log
