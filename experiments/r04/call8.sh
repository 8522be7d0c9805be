#!/bin/bash
# kernel trace of the running pipeline (200 steps), analysed for who waits for whom (experiments/r04/dep_trace.py), gcn and concat
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c8; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in gcn:4 gcn:8 concat:4; do
  depth=${mode#*:}; mode=${mode%:*}; tag=${mode}_d$depth
  rm -rf $O/trace_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$tag -- python3 $R/bench.py --mode $mode --depth $depth --steps 200 --warmup 20 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > $O/bench_$tag.json 2> $O/trace_$tag.log || { tail -5 $O/trace_$tag.log; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench_$tag.json')); print('$tag under the tracer: %.1f us per forward' % (1e3*d['ms_per_step']))"
  python3 $R/experiments/r04/dep_trace.py $O/trace_$tag $depth | tee $O/dep_$tag.txt
  find $O/trace_$tag -name "*.csv" -size +6M -delete
done
