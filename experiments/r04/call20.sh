#!/bin/bash
# the register file as the shared resource: gather 88 VGPRs x up to 5 waves per SIMD, contraction 168 x 2, layer 2 107 x 2 of a SIMD's 512.  With the gather at
# 2 blocks per CU (2 x 88 = 176) it fits BESIDE a contraction block (336) whatever the launch order -- which the flag hand-offs (no bubbles) need.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c20
run() { name=$1; shift
  for form in long short; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c20/$name.$form.$rep.json 2> gpurun_out/r04c20/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c20/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c20/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c20/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-22s 300-step %5.1f  20-step %5.1f  G in situ %5.1f alone %5.1f  check %s' % ('$name', 1e3*l['ms_per_step'], 1e3*s['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], l['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
}
for rep in 1 2; do
  for fl in 0 1; do for g in 6 3 2; do for d in 224 128; do
    run f${fl}_g${g}_d${d} SAGE_PIPE_FLAGS=$fl SAGE_G_PER_CU=$g SAGE_DENSE_BLOCKS=$d
  done; done; done
done 2>&1 | tee gpurun_out/r04c20/ab.log
