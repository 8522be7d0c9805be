#!/bin/bash
# POTENTIAL, not a product: a contraction that never waits for its tile rows (every tile re-reads the block's first tile: cache hits, wrong results), with
# events and with flag hand-offs (no bubbles): what would deeper tile prefetch in the contraction be worth?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c25
run() { name=$1; lib=$2; shift 2
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" SAGE355_LIB=$PWD/experiments/ab/$lib.so timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c25/$name.$form.$rep.json 2> gpurun_out/r04c25/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c25/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c25/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c25/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-18s 300-step %5.1f  20-step %5.1f  G in situ %5.1f' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms']), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
}
for rep in 1 2; do
  run real_events real SAGE_PIPE_FLAGS=0
  run real_flags real SAGE_PIPE_FLAGS=1
  run fakeD_events fakeD SAGE_PIPE_FLAGS=0
  run fakeD_flags fakeD SAGE_PIPE_FLAGS=1
  run fakeD_flags_d6 fakeD SAGE_PIPE_FLAGS=1 SAGE_DEPTH=6
  run fakeD_events_g256 fakeD SAGE_PIPE_FLAGS=0 SAGE_DENSE_BLOCKS=256
done 2>&1 | cut -c1-230 | tee gpurun_out/r04c25/ab.log
