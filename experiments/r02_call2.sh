#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_pipe -- python3 $R/experiments/pipe_sweep.py --steps 100 --warmup 20 --baseline 0 --configs 4:SGDL: 3:SGDD: > $O/trace_pipe.log 2>&1
echo "trace rc=$?"
python3 $R/experiments/pipe_trace.py $O/trace_pipe > $O/trace_pipe.txt 2>&1
cat $O/trace_pipe.txt
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_base -- python3 $R/experiments/pipe_sweep.py --steps 100 --warmup 20 --baseline 1 --configs > $O/trace_base.log 2>&1
python3 $R/experiments/pipe_trace.py $O/trace_base > $O/trace_base.txt 2>&1
cat $O/trace_base.txt
find $O -name "*_kernel_trace.csv" -size +20M -delete
