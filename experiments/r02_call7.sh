#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c7
mkdir -p $O
cd $R
SAGE_G_VARIANT=2 SAGE_G_SLICE_LANES=8 SAGE_G_PER_CU=3 timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q > $O/pytest_v2sl8.log 2>&1; echo "pytest(G rows, SL8) rc=$?"; tail -2 $O/pytest_v2sl8.log
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q > $O/pytest_v2t8.log 2>&1; echo "pytest(G rows, auto SL, trip 8) rc=$?"; tail -2 $O/pytest_v2t8.log
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=2
order=degree
for cfg in "8 16 2" "8 16 3" "8 16 4" "8 8 4" "8 8 6" "16 16 1" "16 16 2" "16 16 3" "16 16 4" "16 8 4"; do
  set -- $cfg
  export SAGE_G_SLICE_LANES=$1 SAGE_G_TRIP=$2 SAGE_G_PER_CU=$3
  tag=${order}_sl$1_t$2_g$3
  CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- $CMD > $O/$tag.log 2>&1
  g=$(python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +G n=" | head -1 | cut -c1-70)
  rm -rf $O/t_$tag
  echo "== $tag | $g"
done
export SAGE_G_SLICE_LANES=8 SAGE_G_TRIP=16 SAGE_G_PER_CU=3
CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p -- $CMD > $O/pmc.log 2>&1
echo "pmc sl8: $(python3 $R/experiments/pmc_gather.py $O/p)"; rm -rf $O/p
