#!/bin/bash
# Express lane for an idle pipeline's batch (all five launches on stream L): tests, then same-box interleaved A/B in both forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c28
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "pipe or pipeline or role" 2>&1 | tail -4 || exit 1
run() { name=$1; shift
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c28/$name.$form.$rep.json 2> gpurun_out/r04c28/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c28/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c28/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c28/$name.short.$rep.json'))
print('rep $rep %-12s 20-step %6.2f  300-step %6.2f  checks %s %s' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], s['timed_path_check']['bit_identical_to_oracle_gated_forward'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2 3; do
  run express SAGE_PIPE_EXPRESS=1
  run off SAGE_PIPE_EXPRESS=0
done 2>&1 | tee gpurun_out/r04c28/ab.log
