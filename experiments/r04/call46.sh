#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c46
timeout -k 10 400 python experiments/r04/two_pipes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c46/two_pipes.txt
