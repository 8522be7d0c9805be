#!/bin/bash
# repeated A/B of forward configurations (3 rounds, interleaved), original node order (the BASELINE workload as generated)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c16
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order ${ORDER:-original} --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag [$ORDER G$SAGE_G_VARIANT T$SAGE_G_TRIP G@$SAGE_G_PER_CU D@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g' | cut -c1-62 | tr '\n' ';')"
}
export SAGE_DENSE_BLOCKS=256
for rep in 1 2 3; do
 for ORDER in original degree; do export ORDER
  SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 one r${rep}_${ORDER}_v0g8 --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
  SAGE_G_VARIANT=1 SAGE_G_PER_CU=6 one r${rep}_${ORDER}_v1g6 --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
  SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=4 one r${rep}_${ORDER}_v2g4 --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
  SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 one r${rep}_${ORDER}_v2g2s --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
  SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=3 SAGE_T16_WAVES=8 SAGE_SO_THREADS=512 one r${rep}_${ORDER}_v2g3s --baseline 1 --bstreams 2 --configs 3:SGDD: 4:SGDL:
 done
done
