#!/bin/bash
# same-box A/B of variant builds: experiments/ab_bench.sh name1 name2 ...   (boxes in the pool differ by ~8 %)
for v in "$@"; do
  for s in 2 1; do
    SAGE355_LIB=$GRAFT_REPO_ROOT/experiments/ab/libsage355_$v.so timeout -k 10 200 python bench.py --streams $s --cpu-seconds 0 --no-parity 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'streams', d['config']['streams_in_flight'], 'us/forward %.1f' % (1e3*d['ms_per_step']), 'stages', d['roofline']['stage_ms'])" || exit 1
  done
done
