#!/bin/bash
# streams by batch parity instead of by role: {S,G} of even / odd batches on streams 0 / 1, {D,L} on streams 2 / 3
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c47
timeout -k 10 300 env SAGE_PIPE_PARITY=1 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "host_threads or bit_identical_to_single" > gpurun_out/r03c47/t.log 2>&1 || { tail -20 gpurun_out/r03c47/t.log; exit 1; }
tail -1 gpurun_out/r03c47/t.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_PIPE_PARITY=0" "SAGE_PIPE_PARITY=1" "SAGE_PIPE_PARITY=1 SAGE_DEPTH=6" "SAGE_PIPE_PARITY=1 SAGE_DEPTH=8" 2>&1 | cut -c1-130 | tee gpurun_out/r03c47/q.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 2 "SAGE_PIPE_PARITY=0" "SAGE_PIPE_PARITY=1" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c47/q.log
