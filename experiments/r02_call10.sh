#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 SAGE_DENSE_VARIANT=0 SAGE_DENSE_BLOCKS=192 SAGE_G_SLICE_LANES=16 SAGE_G_TRIP=8 SAGE_G_PER_CU=2
run() { tag=$1; order=$2; shift; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --baseline 1 --bstreams 1 2 3 4 --order $order --tag $tag --configs > $O/$tag.log 2>&1
  echo "== $tag rc=$? $order [G$SAGE_G_VARIANT SL$SAGE_G_SLICE_LANES T$SAGE_G_TRIP G@$SAGE_G_PER_CU D$SAGE_DENSE_VARIANT@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS]"; grep "us/forward" $O/$tag.log | cut -c1-75
}
run g2 degree
SAGE_G_PER_CU=3 run g3 degree
SAGE_G_PER_CU=4 run g4 degree
SAGE_T16_WAVES=16 SAGE_SO_THREADS=1024 run g2_oldSL degree
SAGE_DENSE_VARIANT=1 SAGE_DENSE_BLOCKS=256 run g2_d1 degree
SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 SAGE_T16_WAVES=16 SAGE_SO_THREADS=1024 run old degree
run g2o original
SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 SAGE_T16_WAVES=16 SAGE_SO_THREADS=1024 run oldo original
