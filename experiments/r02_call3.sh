#!/bin/bash
# G v2 (pipelined rows) alone at several occupancies: single-stream graph replay under the kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "0 8" "1 8" "1 6" "1 4" "1 3" "1 2"; do
  set -- $cfg
  export SAGE_G_VARIANT=$1 SAGE_G_PER_CU=$2
  tag=v$1_g$2
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- python3 $R/experiments/pipe_sweep.py --steps 60 --warmup 20 --baseline 1 --bstreams 1 --configs > $O/$tag.log 2>&1
  echo "== $tag rc=$? $(grep 'us/forward' $O/$tag.log)"
  python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +(So|Si|G|D|L2) n=" | cut -c1-90
  rm -rf $O/t_$tag
done
