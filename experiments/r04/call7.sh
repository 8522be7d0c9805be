#!/bin/bash
# the 512-deep concat contraction as two launches of the lock-step kernel (self chunk beside the gather): tests, then A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c7
timeout -k 10 900 python -m pytest tests/test_gpu_dense_two.py -m gpu -x -q > gpurun_out/r04c7/tests.log 2>&1 || { tail -40 gpurun_out/r04c7/tests.log; exit 1; }
tail -2 gpurun_out/r04c7/tests.log
run() { name=$1; shift
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py --mode concat $a --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c7/$name.$form.$rep.json 2> gpurun_out/r04c7/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c7/$name.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c7/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c7/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-16s 300-step %5.1f  20-step %5.1f  G in situ %5.1f' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms']), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
}
for rep in 1 2 3; do
  run one X=1
  run two SAGE_DENSE_TWO=1
  run two_d6 SAGE_DENSE_TWO=1 SAGE_DEPTH=6
  run one_d6 SAGE_DEPTH=6
  run two_sm SAGE_DENSE_TWO=1 SAGE_TABLE_SLICED=2
  run two_g256 SAGE_DENSE_TWO=1 SAGE_DENSE_BLOCKS=256
done 2>&1 | tee gpurun_out/r04c7/ab.log
