#!/bin/bash
# tail events (a stage's hand-off event as its last kernel's own completion signal, no record packet): pipeline tests, then interleaved A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c40
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "pipe or pipeline or role or express or capture" 2>&1 | tail -4 || exit 1
run() { name=$1; shift
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c40/$name.$form.$rep.json 2> gpurun_out/r04c40/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c40/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c40/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c40/$name.short.$rep.json'))
print('rep $rep %-10s 20-step %6.2f  300-step %6.2f  G in situ %5.1f  checks %s %s' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*l['roofline']['kernel_ms'], s['timed_path_check']['bit_identical_to_oracle_gated_forward'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2 3; do
  run tail SAGE_PIPE_TAIL=1
  run record SAGE_PIPE_TAIL=0
done 2>&1 | tee gpurun_out/r04c40/ab.log
