#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c38
timeout -k 10 400 python experiments/r04/launch_storm.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c38/storm.txt
