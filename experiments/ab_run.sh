#!/bin/bash
# experiments/ab_run.sh <reps> <lib names...>: interleaved same-box A/B of variant builds through the default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/ab; mkdir -p $O
reps=$1; shift
for rep in $(seq $reps); do for name in "$@"; do
  SAGE355_LIB=$PWD/experiments/ab/$name.so timeout -k 10 300 python bench.py --steps ${STEPS:-400} --warmup 50 --cpu-seconds 0 --no-variant --no-parity --scale-variant off $BENCH_ARGS > $O/$name.$rep.json 2> $O/$name.$rep.err || { echo "$name FAILED"; tail -3 $O/$name.$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/$name.$rep.json')); r=d['roofline']
print('%-12s rep $rep: %.1f us/fwd  gather in situ %.1f alone %.1f' % ('$name', 1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone']), {k: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
done; done
