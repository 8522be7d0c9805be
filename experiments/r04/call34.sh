#!/bin/bash
# where does the N = 4 shared-GPU rehearsal stand?  progress marks + a stack dump after 150 s
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c34
export SAGE_BENCH_PROGRESS=$PWD/gpurun_out/r04c34 SAGE_BENCH_STACKS_AFTER=150
s=$(date +%s)
timeout -k 10 330 python bench.py --gpus ${1:-4} --share-device --dist-backend gloo --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r04c34/n.json 2> gpurun_out/r04c34/n.err
echo "rc $? wall $(( $(date +%s) - s )) s"
tail -n 5 gpurun_out/r04c34/rank*.log | cut -c1-200
