#!/bin/bash
# producer / consumer contraction (dense_pc_kernel): the one relaxed test, then same-box A/B SAGE_DENSE_PC=0/1: gcn and concat at configs 3 and 5
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c34
timeout -k 10 300 python -m pytest tests/test_gpu_round2.py -x -q -k "prepared_weight" > gpurun_out/r03c34/tests.log 2>&1 || { tail -30 gpurun_out/r03c34/tests.log; exit 1; }
tail -2 gpurun_out/r03c34/tests.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" 2>&1 | cut -c1-200 | tee gpurun_out/r03c34/gcn.log
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 2 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" "SAGE_DENSE_PC=1 SAGE_TABLE_SLICED=2" 2>&1 | cut -c1-200 | tee gpurun_out/r03c34/concat.log
STEPS=300 BENCH_ARGS="--config 5 --mode concat" bash experiments/env_run.sh 2 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" 2>&1 | cut -c1-200 | tee gpurun_out/r03c34/c5concat.log
STEPS=300 BENCH_ARGS="--config 5" bash experiments/env_run.sh 1 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" 2>&1 | cut -c1-200 | tee gpurun_out/r03c34/c5gcn.log
