#!/bin/bash
# confirm the random search's best combination against the defaults (6 interleaved runs, both forms)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c57
B="SAGE_G_PER_CU=5 SAGE_SO_THREADS=256 SAGE_T16_GRID=768 SAGE_DENSE_BLOCKS=224 SAGE_DEPTH=5"
STEPS=300 bash experiments/env_run.sh 6 "SAGE_X=0" "$B" 2>&1 | cut -c1-60 | tee gpurun_out/r03c57/g.log
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 6 "SAGE_X=0" "$B" 2>&1 | cut -c1-60 | tee -a gpurun_out/r03c57/g.log
