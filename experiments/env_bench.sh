#!/bin/bash
for e in "X=1" "HSA_ENABLE_INTERRUPT=0" "HSA_ENABLE_INTERRUPT=0 HIP_FORCE_DEV_KERNARG=1" "HSA_ENABLE_INTERRUPT=0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
  for s in 1 2 3; do env $e timeout -k 10 200 python bench.py --streams $s --cpu-seconds 0 --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e streams', d['config']['streams_in_flight'], 'us/forward %.1f' % (1e3*d['ms_per_step']))"; done
done
