#!/bin/bash
# just-in-time enqueue: a role's host thread makes its calls only once its producer has finished on the GPU (no wait packet in its queue)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c44
STEPS=300 bash experiments/env_run.sh 2 "SAGE_PIPE_JIT=0" "SAGE_PIPE_JIT=2" "SAGE_PIPE_JIT=6" "SAGE_PIPE_JIT=14" 2>&1 | cut -c1-110 | tee gpurun_out/r03c44/q.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 2 "SAGE_PIPE_JIT=0" "SAGE_PIPE_JIT=2" "SAGE_PIPE_JIT=6" "SAGE_PIPE_JIT=14" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c44/q.log
