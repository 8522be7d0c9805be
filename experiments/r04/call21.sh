#!/bin/bash
# wave issue priority per KERNEL (s_setprio at kernel entry): the gather's / samplers' / layer 2's few instructions ahead of the contraction's many
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c21
STEPS=300 bash experiments/ab_run.sh 2 base pG3 pG1 pGSL3 pSL3 pD3 pG2SL3 pG3D1 2>&1 | cut -c1-230 | tee gpurun_out/r04c21/ab.log
for rep in 1 2; do for name in base pG3 pGSL3 pSL3 pG3D1; do
  SAGE355_LIB=$PWD/experiments/ab/$name.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c21/s_$name.$rep.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r04c21/s_$name.$rep.json')); print('20-step $name rep $rep: %.1f us' % (1e3*d['ms_per_step']))"
done; done 2>&1 | tee gpurun_out/r04c21/short.log
