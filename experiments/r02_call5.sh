#!/bin/bash
# gather alone (serial chain): 128-B vs 256-B slices with the pipelined kernel, original vs degree-sorted ids: time + L2 counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=1
for order in original degree; do
for cfg in "16 6" "16 8" "8 4" "8 6" "8 8"; do
  set -- $cfg
  export SAGE_G_SLICE_LANES=$1 SAGE_G_PER_CU=$2
  tag=${order}_sl$1_g$2
  CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- $CMD > $O/$tag.log 2>&1
  g=$(python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +G n=" | head -1 | cut -c1-70)
  rm -rf $O/t_$tag
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p_$tag -- $CMD > $O/${tag}_pmc.log 2>&1
  c=$(python3 $R/experiments/pmc_gather.py $O/p_$tag)
  rm -rf $O/p_$tag
  echo "== $tag | $g | $c"
done
done
