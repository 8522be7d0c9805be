#!/bin/bash
# r03 call 31: the inner sampling hop on a stream of its own (roles SIGDL): bit-identity tests, then A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c31; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "role_pipeline" > $O/tests.log 2>&1; tail -3 $O/tests.log
grep -q "passed" $O/tests.log || exit 1
STEPS=300 bash experiments/env_run.sh 3 "SAGE_ROLES=SGDL" "SAGE_ROLES=SIGDL" "SAGE_ROLES=SIGDL SAGE_SO_THREADS=1024" "SAGE_ROLES=SIGDL SAGE_DEPTH=6" 2>&1 | cut -c1-110 | tee $O/log.txt
STEPS=20 BENCH_ARGS="--warmup 5" bash experiments/env_run.sh 3 "SAGE_ROLES=SGDL" "SAGE_ROLES=SIGDL" 2>&1 | cut -c1-110 | tee -a $O/log.txt
