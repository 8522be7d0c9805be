#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c29
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SWEEP_GRAPH=1
SAGE_PIPE_SYSFENCE=1 timeout -k 10 300 python3 -X faulthandler $R/experiments/pipe_sweep.py --steps 20 --warmup 5 --order degree --baseline 0 --configs 4:SGDL: > $O/g20.log 2>&1; echo rc=$?
grep -E "us/forward|graph|File|Fatal" $O/g20.log | sed -E 's/ +/ /g' | head -20
