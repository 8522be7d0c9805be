#!/bin/bash
# the gather launched through hipExtLaunchKernelGGL: 1 = its completion signal is the hand-off event (no marker packet);
# 3 = + hipExtAnyOrderLaunch (may start before the previous gather of its stream has drained; enqueued only once its samplers finished)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c45
timeout -k 10 300 env SAGE_PIPE_G_EXT=3 python -m pytest tests/test_gpu_round3.py -x -q -k "host_threads" > gpurun_out/r03c45/t.log 2>&1 || { tail -20 gpurun_out/r03c45/t.log; exit 1; }
tail -1 gpurun_out/r03c45/t.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_PIPE_G_EXT=0" "SAGE_PIPE_G_EXT=1" "SAGE_PIPE_G_EXT=3" 2>&1 | cut -c1-110 | tee gpurun_out/r03c45/q.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 2 "SAGE_PIPE_G_EXT=0" "SAGE_PIPE_G_EXT=1" "SAGE_PIPE_G_EXT=3" 2>&1 | cut -c1-110 | tee -a gpurun_out/r03c45/q.log
