#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_G_VARIANT=1
order=degree
for cfg in "8 2 4" "8 2 6" "8 2 8" "8 4 3" "8 4 4" "8 4 5" "16 2 4" "16 2 3"; do
  set -- $cfg
  export SAGE_G_SLICE_LANES=$1 SAGE_G_ROWS=$2 SAGE_G_PER_CU=$3
  tag=${order}_sl$1_r$2_g$3
  CMD="python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order $order --configs"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t_$tag -- $CMD > $O/$tag.log 2>&1
  g=$(python3 $R/experiments/pipe_trace.py $O/t_$tag 2>&1 | grep -E "^  +G n=" | head -1 | cut -c1-70)
  rm -rf $O/t_$tag
  echo "== $tag | $g"
done
# pipelines, degree order: small-footprint kernels + the candidates above
export SAGE_DENSE_VARIANT=1 SAGE_DENSE_BLOCKS=256 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256
run() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 200 --warmup 20 --baseline 0 --order degree --tag $tag --configs "$@" > $O/$tag.log 2>&1
  echo "== $tag rc=$?  [SL$SAGE_G_SLICE_LANES R$SAGE_G_ROWS G@$SAGE_G_PER_CU D$SAGE_DENSE_VARIANT@$SAGE_DENSE_BLOCKS]"; grep "us/forward" $O/$tag.log | cut -c1-75
}
SAGE_G_SLICE_LANES=16 SAGE_G_ROWS=1 SAGE_G_PER_CU=4 run P16r1g4 4:SGDL: 3:SGDD:
SAGE_G_SLICE_LANES=8 SAGE_G_ROWS=2 SAGE_G_PER_CU=4 run P8r2g4 4:SGDL:
SAGE_G_SLICE_LANES=8 SAGE_G_ROWS=2 SAGE_G_PER_CU=6 run P8r2g6 4:SGDL:
SAGE_G_SLICE_LANES=8 SAGE_G_ROWS=4 SAGE_G_PER_CU=4 run P8r4g4 4:SGDL: 3:SGDD:
SAGE_G_SLICE_LANES=8 SAGE_G_ROWS=4 SAGE_G_PER_CU=3 run P8r4g3 4:SGDL:
