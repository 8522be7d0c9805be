#!/bin/bash
# timeline of the driver's form with the express lane on / off
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c29; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  export SAGE_PIPE_EXPRESS=$mode
  rm -rf $O/trace$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace$mode -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > $O/bench$mode.json 2> $O/trace$mode.log || { tail -5 $O/trace$mode.log; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/bench$mode.json')); print('express=$mode under the tracer: %.1f us per step' % (1e3*d['ms_per_step']))"
  python3 $R/experiments/r04/timeline_trace.py $O/trace$mode 25 > $O/timeline$mode.txt
  sed -n 1,14p $O/timeline$mode.txt
  find $O/trace$mode -name "*.csv" -size +6M -delete
done
