#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c25
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_T16_WAVES=8
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/experiments/pipe_sweep.py --steps 150 --warmup 30 --baseline 0 --order degree --configs 4:SGDL: > $O/trace.log 2>&1
python3 $R/experiments/pipe_trace.py $O/trace > $O/trace.txt 2>&1
grep -E "phase|^  +(So|Si|G|D|L2) n=|gap|running" $O/trace.txt | cut -c1-160
sed -n '/phase/,$p' $O/trace.txt | grep -E "^ +[0-9]" | head -45
rm -rf $O/trace
