#!/bin/bash
# Does the contraction's residency matter?  In-pipeline kernel durations (rocprofv3 --kernel-trace --stats) of the real library and of the
# build whose contraction never waits for its rows (fake_loads.patch, wrong results).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04c26; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SAGE_PIPE_FLAGS=0
for lib in real fakeD; do
  export SAGE355_LIB=$R/experiments/ab/$lib.so
  rm -rf $O/$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$lib -- python3 $R/bench.py --steps 300 --warmup 50 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > $O/$lib.json 2> $O/$lib.log || { echo "$lib failed"; tail -3 $O/$lib.log; exit 1; }
  f=$(find $O/$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib: us per forward $(python3 -c "import json; print(1e3*json.load(open('$O/$lib.json'))['ms_per_step'])")"
  cut -d, -f1-5 $f | cut -c1-150 | sed -n 1,8p
  cp $f $O/${lib}_kernel_stats.csv
  find $O/$lib -name "*_kernel_trace.csv" -delete
done
