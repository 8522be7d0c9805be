#!/bin/bash
# what the driver does at round end: build check, smoke(), the GPU suite, bench.py as it runs it
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c18
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c18/tests.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r04c18/tests.log
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04c18/bench.json 2> gpurun_out/r04c18/bench.err; echo "bench rc=$? in $(( $(date +%s) - t0 )) s"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04c18/bench.json')); v=d['config']['variants']['configs3_rmat23']
print('value %.4g ms/step %.5f fwd_frac %.4f roofline.frac %.4f | configs[3] %.5f fwd_frac %.4f | cpu %s' % (d['value'], d['ms_per_step'], d['roofline']['forward_frac'], d['roofline']['frac'], v['ms_per_step'], v['forward_frac'], d['cpu_baseline']['value']))
PY
