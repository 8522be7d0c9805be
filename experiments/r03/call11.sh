#!/bin/bash
# r03 call 11: the 20-step (driver) form: trace of the timed region; submit_many / depth variants
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=$PWD/gpurun_out/r03c11; mkdir -p $O; R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t20 -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity --preheat-seconds 0.05 > $O/t20.json 2> $O/t20.log
cd $R; python3 experiments/r03/trace20.py $O/t20 20 | head -70
for rep in 1 2 3; do for v in "SAGE_DEPTH=4" "SAGE_DEPTH=5" "SAGE_DEPTH=6" "SAGE_DEPTH=8"; do
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > $O/d.json 2> $O/d.err || tail -3 $O/d.err
  python3 -c "
import json; d=json.load(open('$O/d.json')); print('$v rep $rep: %.2f us/fwd host %.1f' % (1e3*d['ms_per_step'], 1e3*d['config']['host_enqueue_ms_per_step']))"
done; done
find $O -name "*_kernel_trace.csv" -size +8M -delete
