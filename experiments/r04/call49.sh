#!/bin/bash
# the contraction's software-pipelined tile loop against the lock-step loop, same box, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c49
run() { name=$1; lib=$2
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    SAGE355_LIB=$PWD/experiments/ab/$lib.so timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c49/$name.$form.$rep.json 2> gpurun_out/r04c49/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c49/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c49/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c49/$name.short.$rep.json')); r=l['roofline']
print('rep $rep %-10s 20-step %6.2f  300-step %6.2f  G in situ %5.1f  contraction alone %5.1f' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['stage_ms_alone']['layer1_contract']))"
}
for rep in 1 2 3 4; do
  run lockstep lockstep
  run pipelined pipelined
done 2>&1 | tee gpurun_out/r04c49/ab.log
for c in 4; do for lib in lockstep pipelined; do
  SAGE355_LIB=$PWD/experiments/ab/$lib.so timeout -k 10 300 python bench.py --config $c --steps 200 --warmup 30 --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c49/c$c.$lib.json 2>/dev/null && python3 -c "
import json; l=json.load(open('gpurun_out/r04c49/c$c.$lib.json')); print('config $c $lib: %.2f us' % (1e3*l['ms_per_step']))"
done; done 2>&1 | tee -a gpurun_out/r04c49/ab.log
