#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c24
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag [G@$SAGE_G_PER_CU D@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g; s/us\/forward \(submit_many\)/many/; s/\(submit each\) host enqueue/each, host/; s/identical=True//' | tr '\n' ';')"
}
one warm --configs 4:SGDL:
one base --configs 4:SGDL: 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 one t8 --configs 4:SGDL: 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=192 one t8d192 --configs 4:SGDL: 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=128 one t8d128 --configs 4:SGDL: 6:SGDL: 8:SGDL:
SAGE_DENSE_BLOCKS=128 one d128 --configs 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=192 SAGE_G_PER_CU=8 one t8d192g8 --configs 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=192 SAGE_G_PER_CU=7 one t8d192g7 --configs 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=160 one t8d160 --configs 6:SGDL: 8:SGDL:
SAGE_T16_WAVES=8 SAGE_DENSE_BLOCKS=192 SAGE_T16_GRID=256 one t8d192grid256 --configs 6:SGDL: 8:SGDL:
