#!/bin/bash
# r03 call 28: config 5 (400-byte rows) with a zero-padded slice-major copy: 2 slices of 64 floats / 4 of 32 / row-major (one 32-lane slice)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c28; mkdir -p $O
for rep in 1 2 3; do for v in "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICE_FLOATS=64" "SAGE_TABLE_SLICE_FLOATS=32" "SAGE_TABLE_SLICE_FLOATS=64 SAGE_G_VARIANT_SM=1"; do
  env $v timeout -k 10 400 python bench.py --config 5 --steps 100 --cpu-seconds 0 --no-variant > $O/d.json 2> $O/d.err || { echo "$v FAILED"; tail -3 $O/d.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-46s rep $rep: %.2f us  fwd_frac %.3f  G in situ %.1f alone %.1f parity %.1e' % ('$v', 1e3*d['ms_per_step'], r['forward_frac'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], d['parity_max_err_vs_fp64_oracle']))"
done; done | tee $O/log.txt
