#!/bin/bash
# the round's evidence, final state: bench lines + rocprofv3 kernel stats + PMC (profiles/collect.sh), the other configurations' lines
# (experiments/matrix.sh) and their kernel stats (profiles/collect_matrix_stats.sh)
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash profiles/collect.sh r03 > gpurun_out/r03_collect.log 2>&1; tail -25 gpurun_out/r03_collect.log
bash experiments/matrix.sh 100 r03 2>&1 | cut -c1-250 | tee gpurun_out/r03_matrix.log
bash profiles/collect_matrix_stats.sh r03 > gpurun_out/r03_mstats.log 2>&1; tail -12 gpurun_out/r03_mstats.log
