#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c48
SAGE355_LIB=$PWD/experiments/ab/stamps.so timeout -k 10 300 python experiments/r04/dense_stamps_alone.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c48/stamps.txt
