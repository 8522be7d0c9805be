#!/bin/bash
# the cost of a stream-to-stream hand-off by primitive (experiments/r04/handoff.hip)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c36
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 experiments/r04/handoff.hip -o experiments/r04/handoff || exit 1
for k in 20 50; do timeout -k 10 120 ./experiments/r04/handoff $k 200 256 2>&1 | tee -a gpurun_out/r04c36/handoff.txt || exit 1; done
timeout -k 10 120 ./experiments/r04/handoff 20 200 2048 2>&1 | tee -a gpurun_out/r04c36/handoff.txt
