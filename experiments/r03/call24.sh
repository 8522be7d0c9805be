#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c24; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round2.py -x -q -m gpu -k "concat or config5 or wide_and_odd" > $O/tests.log 2>&1; tail -4 $O/tests.log
grep -q "passed" $O/tests.log || exit 1
run(){ cfg="$1"; shift; for rep in 1 2 3; do for v in "$@"; do
  env $v timeout -k 10 400 python bench.py --steps 100 --cpu-seconds 0 --no-variant $cfg > $O/d.json 2> $O/d.err || { echo "$cfg $v FAILED"; tail -3 $O/d.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-28s %-30s rep $rep: %.2f us  fwd_frac %.3f  G in situ %.1f alone %.1f D alone %.1f parity %.1e' % ('$cfg', '$v', 1e3*d['ms_per_step'], r['forward_frac'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], 1e3*r['stage_ms_alone']['layer1_contract'], d['parity_max_err_vs_fp64_oracle']))"
done; done; }
run "--config 3 --mode concat" "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1" "SAGE_TABLE_SLICE_FLOATS=64"
