#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS=300 bash experiments/ab_run.sh 4 agg_sm agg_rm 2>&1 | tee gpurun_out/r03c27.log
