#!/bin/bash
# EXPERIMENT: the gather of batch b+1 waits for the START of the contraction of batch b (the contraction's blocks are resident before the gather floods the CUs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c45
export SAGE355_LIB=$PWD/experiments/ab/dfirst.so
SAGE_PIPE_DFIRST=1 timeout -k 10 300 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round3.py -x -q -m gpu -k "pipe or pipeline or express" 2>&1 | tail -3
run() { name=$1; shift
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c45/$name.$form.$rep.json 2> gpurun_out/r04c45/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c45/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c45/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c45/$name.short.$rep.json'))
print('rep $rep %-10s 20-step %6.2f  300-step %6.2f  G in situ %5.1f  checks %s %s' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*l['roofline']['kernel_ms'], s['timed_path_check']['bit_identical_to_oracle_gated_forward'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2 3; do
  run base SAGE_PIPE_DFIRST=0
  run dfirst SAGE_PIPE_DFIRST=1
  run dfirst_d6 SAGE_PIPE_DFIRST=1 SAGE_DEPTH=6
done 2>&1 | tee gpurun_out/r04c45/ab.log
