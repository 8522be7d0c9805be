#!/bin/bash
for p in ${PIPES:-0 1 2}; do for s in 1 2 3; do
  timeout -k 10 300 python bench.py --pipeline $p --streams $s --warmup 24 --steps 240 --cpu-seconds 0 --no-parity 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pipeline', d['config']['pipelined_sampling'], 'streams', d['config']['streams_in_flight'], 'us/forward %.1f' % (1e3*d['ms_per_step']), 'emb/s %.3g' % d['value'])" || exit 1
done; done
