#!/bin/bash
# dense_pc_kernel, second form (4 full-K consumers + 4 producers, no LDS atomics): stamps, contraction tests, A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c36
SAGE355_LIB=$PWD/experiments/ab/pc_stamps.so timeout -k 10 300 python experiments/r03/pc_stamps.py > gpurun_out/r03c36/stamps.log 2>&1; grep -v amdgpu.ids gpurun_out/r03c36/stamps.log
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_ops.py -x -q > gpurun_out/r03c36/tests.log 2>&1 || { tail -30 gpurun_out/r03c36/tests.log; exit 1; }
tail -2 gpurun_out/r03c36/tests.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" 2>&1 | cut -c1-200 | tee gpurun_out/r03c36/gcn.log
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 2 "SAGE_DENSE_PC=0" "SAGE_DENSE_PC=1" "SAGE_DENSE_PC=1 SAGE_TABLE_SLICED=2" 2>&1 | cut -c1-200 | tee gpurun_out/r03c36/concat.log
