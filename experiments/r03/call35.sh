#!/bin/bash
# dense_pc_kernel diagnostics: why is it slower?  producers at s_setprio 3; consumers without MFMAs; producers without the split
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c35
STEPS=200 bash experiments/ab_run.sh 1 pc_base pc_prio pc_nomfma pc_nosplit 2>&1 | cut -c1-220 | tee gpurun_out/r03c35/ab.log
