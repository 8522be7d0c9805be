#!/bin/bash
# final state: the whole GPU suite once more (role streams are now shared process-wide), then the bench lines for profiles/ (200-step form, driver's form x3)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c13 gpurun_out/r04
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04c13/tests.log 2>&1 || { tail -40 gpurun_out/r04c13/tests.log; exit 1; }
tail -2 gpurun_out/r04c13/tests.log
timeout -k 10 400 python bench.py > gpurun_out/r04/bench.json 2> gpurun_out/r04/bench.err; echo "bench rc=$?"
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r04c13/bench_20_5.$i.json 2> gpurun_out/r04c13/bench_20_5.$i.err
done
cp gpurun_out/r04c13/bench_20_5.1.json gpurun_out/r04/bench_20_5.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04/bench.json')); v=d['config']['variants']['configs3_rmat23']
print('200-step: headline %.1f us fwd_frac %.4f frac %.4f | configs[3] %.1f us fwd_frac %.4f gather frac %.4f alone %.4f' % (1e3*d['ms_per_step'], d['roofline']['forward_frac'], d['roofline']['frac'], 1e3*v['ms_per_step'], v['forward_frac'], v['roofline']['frac'], v['roofline']['frac_alone']))
for i in (1,2,3):
    d=json.load(open('gpurun_out/r04c13/bench_20_5.%d.json' % i)); v=d['config']['variants']['configs3_rmat23']
    print('20-step run %d: headline %.1f us fwd_frac %.4f | configs[3] %.1f us fwd_frac %.4f' % (i, 1e3*d['ms_per_step'], d['roofline']['forward_frac'], 1e3*v['ms_per_step'], v['forward_frac']))
PY
