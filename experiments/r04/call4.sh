#!/bin/bash
# round-4 tests + the bench as the driver runs it (timed: configs[3] now rides in the N = 1 line) + the two-rank rehearsal
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c4
# (first part of this call, green: pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -k "round4 or two_ranks or gpus_2 or ..." -> 16 passed)
t0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04c4/bench_20_5.json 2> gpurun_out/r04c4/bench_20_5.err || { tail -20 gpurun_out/r04c4/bench_20_5.err; exit 1; }
echo "bench.py as the driver runs it: $(( $(date +%s) - t0 )) s wall"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04c4/bench_20_5.json'))
print('headline us/step', 1e3*d['ms_per_step'], 'fwd_frac', d['roofline']['forward_frac'], 'host', d['config']['host'])
v=d['config']['variants']['configs3_rmat23']
print('configs3 us/step', 1e3*v['ms_per_step'], 'fwd_frac', v['forward_frac'], 'parity', v['parity_max_err_vs_fp64_oracle'], v['timed_path_check']['bit_identical_to_oracle_gated_forward'], 'setup s', v['setup_seconds'], 'kernel_ms', v['roofline']['kernel_ms'], v['roofline']['kernel_ms_alone'])
PY
