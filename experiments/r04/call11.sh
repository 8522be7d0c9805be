#!/bin/bash
# the round's evidence on the final state: profiles/collect.sh r04 (bench lines, rocprofv3 stats in the pipeline and alone, PMC passes) + the F1 figures
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c11
bash profiles/collect.sh r04 2>&1 | tee gpurun_out/r04c11/collect.log
timeout -k 10 400 python -m pytest tests/test_gpu_train.py -m gpu -q -s -k "f1_distribution" 2>&1 | grep -E "F1 micro|passed|failed" | tee gpurun_out/r04c11/f1.log
