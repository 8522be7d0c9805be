#!/bin/bash
# config 4 (R-MAT 2^23, rows mostly unique): slice width of the layer-1 gather
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/c4; mkdir -p $O
for sl in 16 32 64; do for pc in 6 4; do
  SAGE_G_SLICE_LANES=$sl SAGE_G_PER_CU=$pc timeout -k 10 300 python bench.py --config 4 --steps 100 --cpu-seconds 0 --no-variant > $O/sl${sl}_pc$pc.json 2> $O/sl${sl}_pc$pc.err || { echo FAILED sl $sl; tail -3 $O/sl${sl}_pc$pc.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/sl${sl}_pc$pc.json')); r=d['roofline']
print('slice lanes $sl per-cu $pc: us/fwd %.1f' % (1e3*d['ms_per_step']), 'parity %.1e' % d['parity_max_err_vs_fp64_oracle'], 'gather alone %.1f in situ %.1f' % (1e3*r['kernel_ms_alone'], 1e3*r['kernel_ms']), r['kernel'])"
done; done
