#!/bin/bash
# two-stream number only, variants interleaved and repeated: experiments/ab_bench2.sh REPS name...
R=$1; shift
for i in $(seq $R); do for v in "$@"; do
  SAGE355_LIB=$GRAFT_REPO_ROOT/experiments/ab/libsage355_$v.so timeout -k 10 200 python bench.py --streams 2 --cpu-seconds 0 --no-parity 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', '%.1f' % (1e3*d['ms_per_step']))" || exit 1
done; done
