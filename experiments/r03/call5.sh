#!/bin/bash
# r03 call 5: training-path tests with the reproducible backward; step time at config-3 size
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c5; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_round3.py tests/test_gpu_engine_train.py tests/test_gpu_backward.py tests/test_gpu_train.py -x -q -m gpu -s -k "not config3_size_against" > $O/tests.log 2>&1; tail -30 $O/tests.log
timeout -k 10 300 python experiments/train_big.py > $O/train_big.log 2>&1; tail -5 $O/train_big.log
