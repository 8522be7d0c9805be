#!/bin/bash
# a gather that FITS beside a contraction block: 8 / 4 neighbours per trip (56 / 42 VGPRs) x 3 blocks per CU = 168 / 144 of the 176 VGPRs a contraction block leaves per SIMD
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c22
timeout -k 10 300 env SAGE_G_TRIP=4 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round2.py -m gpu -x -q -k "golden or config3 or baseline or pipeline_is_bit" > gpurun_out/r04c22/tests.log 2>&1; tail -2 gpurun_out/r04c22/tests.log
STEPS=300 bash experiments/env_run.sh 2 "SAGE_G_TRIP=16 SAGE_G_PER_CU=6" "SAGE_G_TRIP=4 SAGE_G_PER_CU=3" "SAGE_G_TRIP=4 SAGE_G_PER_CU=4" "SAGE_G_TRIP=4 SAGE_G_PER_CU=6" "SAGE_G_TRIP=4 SAGE_G_PER_CU=8" "SAGE_G_TRIP=8 SAGE_G_PER_CU=3" "SAGE_G_TRIP=8 SAGE_G_PER_CU=4" "SAGE_G_TRIP=8 SAGE_G_PER_CU=8" 2>&1 | cut -c1-200 | tee gpurun_out/r04c22/ab.log
