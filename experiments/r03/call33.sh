#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c33
timeout -k 10 300 python experiments/r03/timeline20.py 20 5 4 > gpurun_out/r03c33/t20.log 2>&1 || { tail -20 gpurun_out/r03c33/t20.log; exit 1; }
cat gpurun_out/r03c33/t20.log | grep -v amdgpu.ids
