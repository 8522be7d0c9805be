#!/bin/bash
# r03 call 3: capture patterns 13-17; the training-path tests (engine autograd node)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c3; mkdir -p $O
PATTERNS="13 14 15 16 17" bash experiments/r03/capture_repro.sh > $O/capture.log 2>&1; grep -E "pattern|exit|OK|WRONG|tail nodes|->" $O/capture.log
echo "=== tests"; timeout -k 10 1100 python -m pytest tests/test_gpu_round3.py tests/test_gpu_engine_train.py tests/test_gpu_train.py tests/test_gpu_backward.py -x -q -m gpu -s -k "not config3_size_against" > $O/tests.log 2>&1; tail -25 $O/tests.log
