#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c13
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
order=degree
for cfg in "0 192" "0 256"; do
  set -- $cfg
  export SAGE_DENSE_VARIANT=$1 SAGE_DENSE_BLOCKS=$2
  tag=d$1_b$2
  CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- $CMD > $O/$tag.log 2>&1
  g=$(python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +(D|G) n=" | cut -c1-66 | tr '\n' '|')
  rm -rf $O/t_$tag
  echo "== $tag | $g"
done
