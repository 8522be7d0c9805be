#!/bin/bash
# which resource of a contraction-shaped kernel starves a gather-shaped one?  (experiments/r04/corun.hip / corun.py)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c24
timeout -k 10 400 python experiments/r04/corun.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c24/corun.log
