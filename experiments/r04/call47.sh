#!/bin/bash
# software-pipelined tile loop of the contraction: correctness (contraction / engine / pipeline tests), then alone + in the pipeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c47
set -o pipefail
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "contract or dense or prepared or two_hop or config or pipeline or adversar or nan or huge or split" 2>&1 | tail -5 || exit 1
for rep in 1 2; do
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c47/p.$form.$rep.json 2> gpurun_out/r04c47/p.$form.$rep.err || { tail -3 gpurun_out/r04c47/p.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c47/p.long.$rep.json')); s=json.load(open('gpurun_out/r04c47/p.short.$rep.json')); r=l['roofline']
print('rep $rep pipelined contraction: 20-step %6.2f  300-step %6.2f  G in situ %5.1f  parity %.1e' % (1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*r['kernel_ms'], l['parity_max_err_vs_fp64_oracle']), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
done 2>&1 | tee gpurun_out/r04c47/ab.log
