#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c33
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag rc=$? [kev=$SAGE_PIPE_KERNEL_EVENTS sysfence=$SAGE_PIPE_SYSFENCE] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g; s/us\/forward \(submit_many\)/many/; s/\(submit each\) host enqueue/each, host/' | tr '\n' ';')"
  grep -iE "error|Traceback" $O/$tag.log | head -3
}
one base0 --configs 4:SGDL:
SAGE_PIPE_KERNEL_EVENTS=1 one kev1 --configs 4:SGDL:
SAGE_PIPE_KERNEL_EVENTS=1 SAGE_PIPE_SYSFENCE=1 one kev1sf --configs 4:SGDL:
one base0b --configs 4:SGDL:
SAGE_PIPE_KERNEL_EVENTS=1 one kev1b --configs 4:SGDL: 4:SGDD:
