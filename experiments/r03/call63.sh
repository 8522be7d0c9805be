#!/bin/bash
# concat + slice-major gather once more, now on the 224-block contraction grid
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c63
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 3 "SAGE_TABLE_SLICED=1" "SAGE_TABLE_SLICED=2" "SAGE_TABLE_SLICED=2 SAGE_DENSE_BLOCKS=192" "SAGE_TABLE_SLICED=2 SAGE_G_PER_CU=5" 2>&1 | cut -c1-110 | tee gpurun_out/r03c63/c3.log
