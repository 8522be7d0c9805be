#!/bin/bash
# host enqueue threads: tests, then same-box interleaved A/B of threads off / on / on + run-ahead window, 300-step and 20-step forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c32
timeout -k 10 600 python -m pytest tests/test_gpu_round3.py -x -q -k "host_threads or config3_size" > gpurun_out/r03c32/tests.log 2>&1 || { tail -30 gpurun_out/r03c32/tests.log; exit 1; }
tail -3 gpurun_out/r03c32/tests.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_BENCH_THREADS=0" "SAGE_BENCH_THREADS=1" "SAGE_BENCH_THREADS=1 SAGE_PIPE_WINDOW=6" "SAGE_BENCH_THREADS=1 SAGE_PIPE_WINDOW=10" 2>&1 | cut -c1-110 | tee gpurun_out/r03c32/ab.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 3 "SAGE_BENCH_THREADS=0" "SAGE_BENCH_THREADS=1" "SAGE_BENCH_THREADS=1 SAGE_PIPE_WINDOW=6" "SAGE_BENCH_THREADS=1 SAGE_PIPE_WINDOW=10" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c32/ab.log
