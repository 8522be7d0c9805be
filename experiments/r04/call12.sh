#!/bin/bash
# configs[3] alone against configs[3] as the second workload of bench.py's process (role streams reused since this call); then the other configurations
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c12
for rep in 1 2; do
  timeout -k 10 300 python bench.py --config 4 --steps 200 --warmup 20 --cpu-seconds 0 --no-variant > gpurun_out/r04c12/c4_alone.$rep.json 2> gpurun_out/r04c12/c4_alone.$rep.err
  timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-variant > gpurun_out/r04c12/both.$rep.json 2> gpurun_out/r04c12/both.$rep.err
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant > gpurun_out/r04c12/both20.$rep.json 2> gpurun_out/r04c12/both20.$rep.err
  python3 - <<PY
import json
a=json.load(open('gpurun_out/r04c12/c4_alone.$rep.json')); b=json.load(open('gpurun_out/r04c12/both.$rep.json')); c=json.load(open('gpurun_out/r04c12/both20.$rep.json'))
vb=b['config']['variants']['configs3_rmat23']; vc=c['config']['variants']['configs3_rmat23']
print('rep $rep: configs[3] alone %.1f us (fwd_frac %.3f) | second workload, 200 steps: headline %.1f, configs[3] %.1f (%.3f) | 20 steps: headline %.1f, configs[3] %.1f (%.3f)' % (1e3*a['ms_per_step'], a['roofline']['forward_frac'], 1e3*b['ms_per_step'], 1e3*vb['ms_per_step'], vb['forward_frac'], 1e3*c['ms_per_step'], 1e3*vc['ms_per_step'], vc['forward_frac']))
PY
done 2>&1 | tee gpurun_out/r04c12/c4.log
bash experiments/matrix.sh 100 r04 2>&1 | cut -c1-260 | tee gpurun_out/r04c12/matrix.log
