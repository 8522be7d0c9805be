#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c36
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag rc=$? $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g; s/us\/forward \(submit_many\)/many/; s/\(submit each\) host enqueue/each, host/; s/identical=True//' | tr '\n' ';')"
}
one a --configs 4:SGGL:
one b --configs 4:SGDL:
one c --configs 4:SGGS:
one d --configs 3:SGGL:
one e --configs 4:SGDS:
one f --configs 6:SGGL:
