#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c9
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256
run() { tag=$1; order=$2; shift; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 200 --warmup 20 --baseline 0 --order $order --tag $tag --configs "$@" > $O/$tag.log 2>&1
  echo "== $tag rc=$? $order [SL$SAGE_G_SLICE_LANES T$SAGE_G_TRIP G@$SAGE_G_PER_CU D$SAGE_DENSE_VARIANT@$SAGE_DENSE_BLOCKS So$SAGE_SO_THREADS]"; grep "us/forward" $O/$tag.log | cut -c1-75
}
export SAGE_DENSE_VARIANT=0 SAGE_DENSE_BLOCKS=192 SAGE_G_SLICE_LANES=16
SAGE_G_TRIP=8 SAGE_G_PER_CU=2 run a1 degree 4:SGDL: 3:SGDD:
SAGE_G_TRIP=8 SAGE_G_PER_CU=3 run a2 degree 4:SGDL:
SAGE_G_TRIP=16 SAGE_G_PER_CU=2 run a3 degree 4:SGDL:
SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_DENSE_BLOCKS=256 run a4 degree 4:SGDL:
SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_DENSE_VARIANT=1 SAGE_DENSE_BLOCKS=256 run a5 degree 4:SGDL:
SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_SO_THREADS=512 run a6 degree 4:SGDL:
SAGE_G_SLICE_LANES=8 SAGE_G_TRIP=8 SAGE_G_PER_CU=3 run b1 degree 4:SGDL: 3:SGDD:
SAGE_G_SLICE_LANES=8 SAGE_G_TRIP=16 SAGE_G_PER_CU=2 run b2 degree 4:SGDL:
SAGE_G_TRIP=8 SAGE_G_PER_CU=2 run c1 original 4:SGDL: 3:SGDD:
SAGE_G_SLICE_LANES=8 SAGE_G_TRIP=8 SAGE_G_PER_CU=3 run c2 original 4:SGDL:
# timeline of a1
export SAGE_G_TRIP=8 SAGE_G_PER_CU=2
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/experiments/pipe_sweep.py --steps 100 --warmup 20 --baseline 0 --order degree --configs 4:SGDL: > $O/trace.log 2>&1
python3 $R/experiments/pipe_trace.py $O/trace > $O/trace.txt 2>&1
grep -E "phase|^  +(So|Si|G|D|L2) n=|gap|running" $O/trace.txt | cut -c1-150
rm -rf $O/trace
