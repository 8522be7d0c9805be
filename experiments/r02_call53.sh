#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for d in 4 6 8; do
  BENCH_ARGS="--depth $d" bash experiments/env_run.sh 1 "X=0" "SAGE_DENSE_BLOCKS=128" "SAGE_DENSE_BLOCKS=128 SAGE_T16_GRID=256" | sed "s/^/depth $d /"
done
