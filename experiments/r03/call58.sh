#!/bin/bash
# the in-pipeline kernel summary again, WITHOUT the variants' pipelines in the same trace
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r03
mkdir -p "$O"; rm -rf "$O/stats"; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --cpu-seconds 0 --no-variant \
    > "$O/bench_under_rocprof.json" 2> "$O/stats.log"; echo "stats rc=$?"
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
