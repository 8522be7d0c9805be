#!/bin/bash
# just-in-time launches ON TOP of tail events: the critical streams' kernels with nothing between them
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c41
run() { name=$1; shift
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c41/$name.$form.$rep.json 2> gpurun_out/r04c41/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c41/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c41/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c41/$name.short.$rep.json'))
print('rep $rep %-12s 20-step %6.2f  300-step %6.2f  G in situ %5.1f  checks %s %s' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*l['roofline']['kernel_ms'], s['timed_path_check']['bit_identical_to_oracle_gated_forward'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2; do
  run base SAGE_PIPE_JIT=
  run jit_D SAGE_PIPE_JIT=D
  run jit_GD SAGE_PIPE_JIT=GD
  run jit_GDL SAGE_PIPE_JIT=GDL
  run jit_SGDL SAGE_PIPE_JIT=SGDL
  run jit_GD_d6 SAGE_PIPE_JIT=GD SAGE_DEPTH=6
done 2>&1 | tee gpurun_out/r04c41/ab.log
