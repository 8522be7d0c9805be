#!/bin/bash
# hand-off waits whose producer has already finished are not enqueued (hipEventQuery first): SAGE_PIPE_QUERY=0/1
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c43
STEPS=300 bash experiments/env_run.sh 3 "SAGE_PIPE_QUERY=0" "SAGE_PIPE_QUERY=1" 2>&1 | cut -c1-110 | tee gpurun_out/r03c43/q.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 3 "SAGE_PIPE_QUERY=0" "SAGE_PIPE_QUERY=1" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c43/q.log
