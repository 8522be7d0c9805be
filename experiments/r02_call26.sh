#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c26
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_T16_WAVES=8
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g; s/us\/forward \(submit_many\)/many/; s/\(submit each\) host enqueue/each, host/; s/identical=True//' | tr '\n' ';')"
}
one a --configs 4:SGDL:
one b --configs 4:SGDL:G-1
one c --configs 4:SGDL:S-1,D-1,L-1
one d --configs 4:SGDL:S-1
one e --configs 4:SGDL:D-1,L-1
one f --configs 4:SGDL:L-1
