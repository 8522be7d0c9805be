#!/bin/bash
# EXPERIMENT: 64-byte slices taken in two phases (an XCD's L2 holds one slice at a time) against the 128-byte default
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c50
export SAGE355_LIB=$PWD/experiments/ab/slice64.so
run() { name=$1; shift
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c50/$name.$form.$rep.json 2> gpurun_out/r04c50/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c50/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c50/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c50/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-10s 20-step %6.2f  300-step %6.2f  G in situ %5.1f  G alone %5.1f  parity %.1e %s' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['stage_ms_alone']['layer1_gather'], l['parity_max_err_vs_fp64_oracle'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2; do
  run w32 SAGE_TABLE_SLICE_FLOATS=32
  run w16 SAGE_TABLE_SLICE_FLOATS=16
done 2>&1 | tee gpurun_out/r04c50/ab.log
