#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c12; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_engine_train.py tests/test_gpu_backward.py -x -q -m gpu -k "not config3_size_against and not reference_f1 and not drop_in" > $O/tests.log 2>&1; tail -5 $O/tests.log
timeout -k 10 300 python experiments/train_big.py > $O/train_big.log 2>&1; grep -E "ms per|captured|grad max" $O/train_big.log
