#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=2 SAGE_G_SLICE_LANES=16 SAGE_G_TRIP=8 SAGE_G_PER_CU=4
order=degree
for cfg in "0 192 1" "0 256 1" "1 256 1" "1 384 1" "1 512 1" "1 512 0" "1 768 1"; do
  set -- $cfg
  export SAGE_DENSE_VARIANT=$1 SAGE_DENSE_BLOCKS=$2 SAGE_DENSE_PREFETCH=$3
  tag=d$1_b$2_p$3
  CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- $CMD > $O/$tag.log 2>&1
  g=$(python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +(D|G) n=" | cut -c1-66 | tr '\n' '|')
  rm -rf $O/t_$tag
  echo "== $tag | $g"
done
