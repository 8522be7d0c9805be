#!/bin/bash
# the whole GPU suite on the new sampler launch path (SampleArgs, capped inner grid, early id request), then A/B of the early request
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c3
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c3/tests.log 2>&1 || { tail -40 gpurun_out/r04c3/tests.log; exit 1; }
tail -2 gpurun_out/r04c3/tests.log
STEPS=300 bash experiments/ab_run.sh 3 noearly early 2>&1 | cut -c1-230 | tee gpurun_out/r04c3/ab.log
for rep in 1 2 3; do for name in noearly early; do
  SAGE355_LIB=$PWD/experiments/ab/$name.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > gpurun_out/r04c3/s_$name.$rep.json 2>/dev/null
  python3 -c "
import json; d=json.load(open('gpurun_out/r04c3/s_$name.$rep.json')); print('20-step $name rep $rep: %.1f us' % (1e3*d['ms_per_step']))"
done; done 2>&1 | tee gpurun_out/r04c3/short.log
