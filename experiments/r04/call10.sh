#!/bin/bash
# the whole GPU suite on the round's final library (round-3 kernels, ABI 5, pipe fixes), timed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c10
t0=$(date +%s)
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r04c10/tests.log 2>&1; rc=$?
echo "suite: rc=$rc, $(( $(date +%s) - t0 )) s"
tail -25 gpurun_out/r04c10/tests.log
exit $rc
