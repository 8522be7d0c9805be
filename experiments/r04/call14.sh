#!/bin/bash
# one barrier packet less on the gather's stream: role G's host thread waits for the samplers' event itself (SAGE_PIPE_JIT=G), depth 4 / 6 / 8; also G+D, and all roles
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c14
run() { name=$1; shift
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c14/$name.$form.$rep.json 2> gpurun_out/r04c14/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c14/$name.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c14/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c14/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-12s 300-step %5.1f  20-step %5.1f  G in situ %5.1f  check %s' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2 3; do
  run base X=1
  run jitG SAGE_PIPE_JIT=G
  run jitG_d6 SAGE_PIPE_JIT=G SAGE_DEPTH=6
  run jitG_d8 SAGE_PIPE_JIT=G SAGE_DEPTH=8
  run jitGD_d8 SAGE_PIPE_JIT=GD SAGE_DEPTH=8
  run jitGDL_d8 SAGE_PIPE_JIT=GDL SAGE_DEPTH=8
  run base_d8 SAGE_DEPTH=8
done 2>&1 | tee gpurun_out/r04c14/ab.log
