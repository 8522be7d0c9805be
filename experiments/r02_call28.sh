#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c28
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SWEEP_GRAPH=1
timeout -k 10 400 python3 $R/experiments/pipe_sweep.py --steps 200 --warmup 20 --order degree --baseline 0 --configs 4:SGDL: 4:SGDD: 6:SGDL: > $O/g.log 2>&1; echo rc=$?
grep -E "us/forward|graph" $O/g.log | sed -E 's/ +/ /g'
timeout -k 10 400 python3 $R/experiments/pipe_sweep.py --steps 20 --warmup 5 --order degree --baseline 0 --configs 4:SGDL: > $O/g20.log 2>&1; echo rc=$?
grep -E "us/forward|graph" $O/g20.log | sed -E 's/ +/ /g'
