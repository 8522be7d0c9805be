#!/bin/bash
# r03 call 22: slice-major table x gather variant 2 (one destination row per lane group): slice width, blocks per CU, trip, depth
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c22; mkdir -p $O
run(){ for rep in 1 2; do for v in "$@"; do
  env $v timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-variant > $O/d.json 2> $O/d.err || { echo "$v FAILED"; tail -3 $O/d.err; continue; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-78s rep $rep: %.2f us  G in situ %.1f alone %.1f' % ('$v', 1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone']), {k[:8]: round(x*1e3,1) for k,x in r['stage_ms_alone'].items()})"
done; done; }
B="SAGE_TABLE_SLICE_FLOATS=32 SAGE_G_VARIANT=2"
run "$B" "$B SAGE_G_PER_CU=4" "$B SAGE_G_PER_CU=3" "$B SAGE_G_PER_CU=8" "$B SAGE_G_TRIP=8" "$B SAGE_DEPTH=6" "$B SAGE_DENSE_BLOCKS=192" \
    "SAGE_TABLE_SLICE_FLOATS=64 SAGE_G_VARIANT=2" "SAGE_TABLE_SLICE_FLOATS=64 SAGE_G_VARIANT=2 SAGE_G_PER_CU=4" 2>&1 | tee $O/log.txt
