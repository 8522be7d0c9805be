#!/bin/bash
# where do the 20 steps of the driver's form go?  kernel trace of `bench.py --steps 20 --warmup 5`, per-batch timeline of the timed region
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1; do
  rm -rf $O/trace$rep
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$rep -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > $O/bench$rep.json 2> $O/trace$rep.log || { tail -5 $O/trace$rep.log; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench$rep.json')); print('run $rep under the tracer: %.1f us per step' % (1e3*d['ms_per_step']))"
  python3 $R/experiments/r04/timeline_trace.py $O/trace$rep 25 | tee $O/timeline$rep.txt
  find $O/trace$rep -name "*.csv" -size +6M -delete
done
