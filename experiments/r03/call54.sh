#!/bin/bash
# concat encoder with the slice-major gather again (host threads on, lock-step contraction): configs 3 and 5
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c54
STEPS=300 BENCH_ARGS="--mode concat" bash experiments/env_run.sh 2 "SAGE_TABLE_SLICED=1" "SAGE_TABLE_SLICED=2" "SAGE_TABLE_SLICED=2 SAGE_TABLE_SLICE_FLOATS=64" "SAGE_TABLE_SLICED=2 SAGE_DEPTH=6" 2>&1 | cut -c1-200 | tee gpurun_out/r03c54/c3.log
