#!/bin/bash
# express lane, ABI 6: the whole GPU suite, smoke, then the driver's form twice
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
set -o pipefail; timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 &&
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04c30/bench_20_5.$i.json 2> gpurun_out/r04c30/bench_20_5.$i.err && python3 -c "
import json; l=json.load(open('gpurun_out/r04c30/bench_20_5.$i.json')); v=l['config']['variants']['configs3_rmat23']
print('driver form: %.2f us (fwd_frac %.3f, check %s); configs[3] %.2f us (fwd_frac %.3f, check %s)' % (1e3*l['ms_per_step'], l['roofline']['forward_frac'], l['timed_path_check']['bit_identical_to_oracle_gated_forward'], 1e3*v['ms_per_step'], v['forward_frac'], v['timed_path_check']['bit_identical_to_oracle_gated_forward']))"; done
