#!/bin/bash
# r03 call 26: the means (agg1) slice-major too: tests, then A/B against row-major means and against the row-major table
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c26; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/test_gpu_forward.py tests/test_gpu_round2.py tests/test_gpu_round3.py tests/test_gpu_engine_train.py -x -q -m gpu -k "not reference_f1 and not drop_in and not config4 and not config5" > $O/tests.log 2>&1; tail -4 $O/tests.log
grep -q "passed" $O/tests.log || exit 1
for rep in 1 2 3; do for v in "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1"; do for sw in "200 20" "20 5"; do set -- $sw
  env $v timeout -k 10 300 python bench.py --steps $1 --warmup $2 --cpu-seconds 0 --no-variant > $O/d.json 2> $O/d.err || { tail -3 $O/d.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-20s steps %3d rep $rep: %.2f us  G in situ %.1f alone %.1f  D alone %.1f parity %.1e' % ('$v', $1, 1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], 1e3*r['stage_ms_alone']['layer1_contract'], d['parity_max_err_vs_fp64_oracle']))"
done; done; done | tee $O/log.txt
