#!/bin/bash
# the drop-in loop's step-time statistics since the product caps torch's intra-op pool (five runs of the test, printed)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c23
for i in 1 2 3 4 5; do
  timeout -k 10 300 python -m pytest tests/test_gpu_round3.py -m gpu -q -s -k "reference_loop_shape" 2>&1 | grep -E "drop-in loop|passed|failed|sage355: torch"
done | tee gpurun_out/r04c23/dropin.log
