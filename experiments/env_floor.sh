#!/bin/bash
# does a runtime knob change the per-kernel cost inside a replayed graph?  (experiments/mb_floor.py under rocprofv3)
cd /tmp && export TMPDIR=/tmp
run(){ rm -rf /tmp/fl; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/fl -- python3 $GRAFT_REPO_ROOT/experiments/mb_floor.py > /tmp/fl.log 2>&1; echo "== $1"; python3 $GRAFT_REPO_ROOT/experiments/trace_floor.py /tmp/fl | grep -E "empty', '64'|k_write|gap"; }
run default
export HIP_FORCE_DEV_KERNARG=1; run HIP_FORCE_DEV_KERNARG=1; unset HIP_FORCE_DEV_KERNARG
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=1; run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1; unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0; unset DEBUG_CLR_GRAPH_PACKET_CAPTURE
export HSA_ENABLE_INTERRUPT=0; run HSA_ENABLE_INTERRUPT=0; unset HSA_ENABLE_INTERRUPT
cd $GRAFT_REPO_ROOT
for e in "X=1" "HIP_FORCE_DEV_KERNARG=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0"; do
  for s in 1 2; do env $e timeout -k 10 200 python bench.py --streams $s --cpu-seconds 0 --no-parity 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e streams', d['config']['streams_in_flight'], 'us/forward %.1f' % (1e3*d['ms_per_step']))"; done
done
