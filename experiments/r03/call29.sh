#!/bin/bash
# r03 call 29: with the gather no longer the pacemaker (slice-major default), do the other stages' tunables matter now?
cd ${GRAFT_REPO_ROOT:-/root/repo}
STEPS=300 bash experiments/env_run.sh 2 "SAGE_DEPTH=4" "SAGE_T16_WAVES=16" "SAGE_SO_THREADS=512" "SAGE_SAMPLE_FUSED=1" "SAGE_DENSE_BLOCKS=320" "SAGE_DENSE_BLOCKS=384" \
   "SAGE_ROLES=SGDD" "SAGE_ROLES=SGGL" "SAGE_ROLES=SSDL" "SAGE_T16_GRID=256" "SAGE_T16_GRID=1024" 2>&1 | tee gpurun_out/r03c29.log
