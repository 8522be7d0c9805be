#!/bin/bash
# same-box interleaved A/B: round-3 library / round-4 sampler without the strided tail / with it, grid caps 0 / 2048 / 3072 / 4096
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c5
run() { # name lib env...
  name=$1; lib=$2; shift 2
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" SAGE355_LIB=$PWD/experiments/ab/$lib.so timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c5/$name.$form.$rep.json 2> gpurun_out/r04c5/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c5/$name.$form.$rep.err; exit 1; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c5/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c5/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-14s 300-step %5.1f  20-step %5.1f  G in situ %5.1f' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms']), {k[:8]: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"
}
for rep in 1 2 3; do
  run r3 r3 X=1
  run noloop_g0 noloop SAGE_SI_GRID=0
  run r4_g0 r4 SAGE_SI_GRID=0
  run r4_g2048 r4 SAGE_SI_GRID=2048
  run r4_g3072 r4 SAGE_SI_GRID=3072
  run r4_g4096 r4 SAGE_SI_GRID=4096
done 2>&1 | tee gpurun_out/r04c5/ab.log
