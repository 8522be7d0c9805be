#!/bin/bash
# stream priorities for the late stages (older batches first), 20-step and 300-step forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c42
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 3 "SAGE_PIPE_PRIO=" "SAGE_PIPE_PRIO=L" "SAGE_PIPE_PRIO=DL" "SAGE_PIPE_PRIO=GDL" "SAGE_PIPE_PRIO=S" 2>&1 | cut -c1-110 | tee gpurun_out/r03c42/prio.log
STEPS=300 bash experiments/env_run.sh 1 "SAGE_PIPE_PRIO=" "SAGE_PIPE_PRIO=L" "SAGE_PIPE_PRIO=DL" "SAGE_PIPE_PRIO=GDL" "SAGE_PIPE_PRIO=S" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c42/prio.log
