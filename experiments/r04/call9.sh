#!/bin/bash
# flag hand-offs (one-thread signal / gate kernels instead of hipEvent record / wait pairs): pipeline tests, then A/B in both forms, gcn and concat
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c9
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pipe" > gpurun_out/r04c9/tests.log 2>&1 || { tail -40 gpurun_out/r04c9/tests.log; exit 1; }
tail -2 gpurun_out/r04c9/tests.log
run() { name=$1; mode=$2; shift 2
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py --mode $mode $a --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c9/$name.$form.$rep.json 2> gpurun_out/r04c9/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c9/$name.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c9/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c9/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-16s 300-step %5.1f  20-step %5.1f  G in situ %5.1f  host %4.1f' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms'], 1e3*l['config']['host_enqueue_ms_per_step']))"
}
for rep in 1 2 3; do
  run events gcn SAGE_PIPE_FLAGS=0
  run flags gcn SAGE_PIPE_FLAGS=1
  run flags_d6 gcn SAGE_PIPE_FLAGS=1 SAGE_DEPTH=6
  run events_cc concat SAGE_PIPE_FLAGS=0
  run flags_cc concat SAGE_PIPE_FLAGS=1
done 2>&1 | tee gpurun_out/r04c9/ab.log
