#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c15
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; order=$2; shift; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order $order --tag $tag "$@" > $O/$tag.log 2>&1
  echo "== $tag rc=$? $order [G$SAGE_G_VARIANT T$SAGE_G_TRIP G@$SAGE_G_PER_CU D@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS]"; grep "us/forward" $O/$tag.log | cut -c1-75
}
export SAGE_DENSE_BLOCKS=256
SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 run A0 degree --baseline 1 --bstreams 2 --configs 3:SGDD:
SAGE_G_VARIANT=1 SAGE_G_PER_CU=6 run A1 degree --baseline 1 --bstreams 2 --configs 3:SGDD:
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=4 run A2 degree --baseline 1 --bstreams 2 --configs 3:SGDD:
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 run B2 degree --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 SAGE_DENSE_BLOCKS=192 run B3 degree --baseline 0 --configs 3:SGDD:
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=3 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 run B4 degree --baseline 0 --configs 3:SGDD:
SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 run C0 original --baseline 1 --bstreams 2 --configs 3:SGDD:
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 run C2 original --baseline 0 --configs 3:SGDD:
