#!/bin/bash
# r03 call 1: (a) capture patterns, (b) cache policy of the intermediates' stores / loads, (c) contraction grid for the concat encoder
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c1; mkdir -p $O
bash experiments/r03/capture_repro.sh > $O/capture.log 2>&1; tail -40 $O/capture.log
echo "=== store / load policies (gcn, default bench)"; STEPS=400 bash experiments/ab_run.sh 2 base st_sc1 agg_sc1 st_plain all_plain st_sc01 2>&1 | tee $O/ab_policy.log
echo "=== concat: contraction grid"; BENCH_ARGS="--mode concat" STEPS=300 bash experiments/env_run.sh 2 "SAGE_DENSE_BLOCKS=256" "SAGE_DENSE_BLOCKS=224" "SAGE_DENSE_BLOCKS=192" 2>&1 | tee $O/env_concat.log
