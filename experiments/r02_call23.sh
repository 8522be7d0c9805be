#!/bin/bash
# with one HW queue per role stream: re-sweep the launch tunables and the grouped hand-off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c23
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag [G$SAGE_G_VARIANT T$SAGE_G_TRIP G@$SAGE_G_PER_CU D@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g' | cut -c1-50 | tr '\n' ';')"
}
one warm --configs 4:SGDL:
one base --configs 4:SGDL: 8:SGDL:g2 8:SGDL:g4 6:SGDL:g2 6:SGDL:g3 8:SGDL: 4:SGDL:g2
SAGE_G_PER_CU=8 one g8 --configs 4:SGDL: 8:SGDL:g2
SAGE_G_PER_CU=5 one g5 --configs 4:SGDL: 8:SGDL:g2
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=4 one v2g4 --configs 4:SGDL: 8:SGDL:g2
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=3 SAGE_T16_WAVES=8 SAGE_SO_THREADS=512 one v2g3s --configs 4:SGDL: 8:SGDL:g2
SAGE_G_VARIANT=2 SAGE_G_TRIP=8 SAGE_G_PER_CU=2 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256 one v2g2s --configs 4:SGDL: 8:SGDL:g2
SAGE_G_VARIANT=2 SAGE_G_TRIP=16 SAGE_G_PER_CU=3 one v2t16g3 --configs 4:SGDL: 8:SGDL:g2
SAGE_G_VARIANT=0 SAGE_G_PER_CU=8 one v0g8 --configs 4:SGDL: 8:SGDL:g2
SAGE_DENSE_BLOCKS=192 one d192 --configs 4:SGDL: 8:SGDL:g2
SAGE_DENSE_BLOCKS=128 one d128 --configs 4:SGDL: 8:SGDL:g2
SAGE_T16_WAVES=8 one t8 --configs 4:SGDL: 8:SGDL:g2
SAGE_SO_THREADS=512 one so512 --configs 4:SGDL: 8:SGDL:g2
