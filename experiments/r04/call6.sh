#!/bin/bash
# inner hop with 2 / 4 nodes per lane group (a quarter of the waves): tests under SAGE_SI_ROWS=4, then interleaved A/B in both forms
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c6
SAGE_SI_ROWS=4 timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "sampl or pipe or two_hop or config or golden or frontier" > gpurun_out/r04c6/tests.log 2>&1 || { tail -40 gpurun_out/r04c6/tests.log; exit 1; }
tail -2 gpurun_out/r04c6/tests.log
run() { name=$1; shift
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c6/$name.$form.$rep.json 2> gpurun_out/r04c6/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c6/$name.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c6/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c6/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-10s 300-step %5.1f  20-step %5.1f  G in situ %5.1f' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms']), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
}
for rep in 1 2 3; do
  run rows1 SAGE_SI_ROWS=1
  run rows2 SAGE_SI_ROWS=2
  run rows4 SAGE_SI_ROWS=4
done 2>&1 | tee gpurun_out/r04c6/ab.log
