#!/bin/bash
# loaded latency of one dependent global load: idle / beside a copy / beside a gather / beside the library's forward
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c19
timeout -k 10 300 python experiments/r04/loaded_latency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c19/latency.log
