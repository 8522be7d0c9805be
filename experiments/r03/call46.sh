#!/bin/bash
# a second gather stream (odd batches): consecutive gathers may overlap at their boundaries
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c46
timeout -k 10 300 env SAGE_PIPE_G2=1 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round2.py -x -q -k "host_threads or bit_identical_to_single" > gpurun_out/r03c46/t.log 2>&1 || { tail -20 gpurun_out/r03c46/t.log; exit 1; }
tail -1 gpurun_out/r03c46/t.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_PIPE_G2=0" "SAGE_PIPE_G2=1" "SAGE_PIPE_G2=1 GPU_MAX_HW_QUEUES=16" 2>&1 | cut -c1-130 | tee gpurun_out/r03c46/q.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 2 "SAGE_PIPE_G2=0" "SAGE_PIPE_G2=1" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c46/q.log
