#!/bin/bash
for v in "$@"; do for mode in gcn concat; do
  if [ "$v" = main ]; then unset SAGE355_LIB; else export SAGE355_LIB=$GRAFT_REPO_ROOT/experiments/ab/libsage355_$v.so; fi
  timeout -k 10 300 python bench.py --config 5 --mode $mode --steps 100 --cpu-seconds 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v $mode', 'us/fwd %.1f' % (1e3*d['ms_per_step']), 'parity %.1e' % d['parity_max_err_vs_fp64_oracle'], r['kernel'][:20], {k: round(v*1e3,1) for k,v in r['stage_ms'].items()})" || exit 1
done; done
