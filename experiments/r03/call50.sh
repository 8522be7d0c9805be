#!/bin/bash
# config 4 (R-MAT 2^23 / 128 M edges, the scaling workload; 82.9 us = 0.40 with the defaults): slice width, gather blocks per CU, depth
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c50
STEPS=200 BENCH_ARGS="--config 4" bash experiments/env_run.sh 1 "SAGE_X=0" "SAGE_TABLE_SLICE_FLOATS=64" "SAGE_TABLE_SLICE_FLOATS=128" "SAGE_G_PER_CU=8" "SAGE_G_PER_CU=5" "SAGE_DEPTH=6" "SAGE_TABLE_SLICED=0" "SAGE_X=0" 2>&1 | cut -c1-150 | tee gpurun_out/r03c50/c4.log
