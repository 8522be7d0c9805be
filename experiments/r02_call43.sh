#!/bin/bash
# timeline of the default role pipeline: gaps on the gather's queue
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/tr43; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $O/t -o x --output-format csv -- python3 bench.py --steps 400 --warmup 50 --cpu-seconds 0 --no-variant --no-parity > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 experiments/pipe_trace.py $O/t > $O/trace.txt; cat $O/trace.txt | cut -c1-220 | tail -60
rm -rf $O/t
