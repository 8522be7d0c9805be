#!/bin/bash
# r03 call 21: slice-major table with 128-B / 256-B / 512-B slices (SAGE_TABLE_SLICE_FLOATS = 32 / 64 / 128), against row-major
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c21; mkdir -p $O
for rep in 1 2 3; do for v in "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICE_FLOATS=64" "SAGE_TABLE_SLICE_FLOATS=32" "SAGE_TABLE_SLICE_FLOATS=128" "SAGE_TABLE_SLICE_FLOATS=32 SAGE_G_VARIANT=2"; do
  env $v timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-variant > $O/d.json 2> $O/d.err || { echo "$v FAILED"; tail -3 $O/d.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-46s rep $rep: %.2f us  G in situ %.1f alone %.1f  parity %.1e' % ('$v', 1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], d['parity_max_err_vs_fp64_oracle']))"
done; done | tee $O/log.txt
