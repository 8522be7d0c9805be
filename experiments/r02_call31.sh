#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c31
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for so in 512 1024; do
export SAGE_SO_THREADS=$so
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order degree --configs > $O/t.log 2>&1
echo "fused sampler, $so threads:"; python3 $R/experiments/pipe_trace.py $O/t 2>&1 | grep -E "phase|^  +(So|Si|G|D|L2) n=" | cut -c1-100; rm -rf $O/t
done
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag [fused=$SAGE_SAMPLE_FUSED So$SAGE_SO_THREADS] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g; s/us\/forward \(submit_many\)/many/; s/\(submit each\) host enqueue/each, host/; s/identical=True//' | tr '\n' ';')"
}
SAGE_SO_THREADS=512 one f512 --configs 4:SGDL: 4:SGDL:
SAGE_SO_THREADS=1024 one f1024 --configs 4:SGDL: 4:SGDL:
SAGE_SAMPLE_FUSED=0 SAGE_SO_THREADS=1024 one u1024 --configs 4:SGDL: 4:SGDL:
