#!/bin/bash
# pipeline depth now that the host is off the critical path (300-step and 20-step forms)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c39
STEPS=300 bash experiments/env_run.sh 2 "SAGE_DEPTH=4" "SAGE_DEPTH=5" "SAGE_DEPTH=6" "SAGE_DEPTH=8" "SAGE_DEPTH=8 SAGE_PIPE_WINDOW=12" 2>&1 | cut -c1-120 | tee gpurun_out/r03c39/depth.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 2 "SAGE_DEPTH=4" "SAGE_DEPTH=5" "SAGE_DEPTH=6" "SAGE_DEPTH=8" 2>&1 | cut -c1-120 | tee -a gpurun_out/r03c39/depth.log
