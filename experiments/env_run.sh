#!/bin/bash
# experiments/env_run.sh <reps> "VAR=VAL VAR=VAL" ...: interleaved same-box A/B of launch tunables through the default bench
# (LIB=<name> picks experiments/ab/<name>.so for every setting)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/env; mkdir -p $O
reps=$1; shift
[ -n "$LIB" ] && export SAGE355_LIB=$PWD/experiments/ab/$LIB.so
for rep in $(seq $reps); do i=0; for e in "$@"; do i=$((i+1))
  env $e timeout -k 10 300 python bench.py --steps ${STEPS:-400} --warmup 50 --cpu-seconds 0 --no-variant --no-parity --scale-variant off $BENCH_ARGS > $O/e$i.$rep.json 2> $O/e$i.$rep.err || { echo "$e FAILED"; tail -3 $O/e$i.$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/e$i.$rep.json')); r=d['roofline']
print('rep $rep %5.1f us/fwd  G in situ %5.1f alone %5.1f  host %4.1f | %s' % (1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], 1e3*d['config']['host_enqueue_ms_per_step'], '$e'), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
done; done
