#!/bin/bash
# r03 call 23: the slice-major default (128-B slices, rows kernel) against the row-major table on the other configurations
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c23; mkdir -p $O
run(){ cfg="$1"; shift; for rep in 1 2; do for v in "$@"; do
  env $v timeout -k 10 400 python bench.py --steps 100 --cpu-seconds 0 --no-variant $cfg > $O/d.json 2> $O/d.err || { echo "$cfg $v FAILED"; tail -3 $O/d.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-28s %-44s rep $rep: %.2f us  fwd_frac %.3f  G in situ %.1f alone %.1f parity %.1e' % ('$cfg', '$v', 1e3*d['ms_per_step'], r['forward_frac'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], d['parity_max_err_vs_fp64_oracle']))"
done; done; }
run "--config 3 --mode concat" "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1" "SAGE_TABLE_SLICE_FLOATS=64 SAGE_G_VARIANT_SM=1"
run "--config 3 --self-loop" "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1"
run "--config 4" "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1" "SAGE_TABLE_SLICE_FLOATS=64" "SAGE_TABLE_SLICE_FLOATS=64 SAGE_G_VARIANT_SM=1" "SAGE_TABLE_SLICE_FLOATS=128"
run "--config 3 --engine-layout input" "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1"
