#!/bin/bash
# per-kernel durations (rocprofv3 kernel trace, single stream graph replay) of variant builds: trace_variants.sh name ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export SAGE355_LIB=$GRAFT_REPO_ROOT/experiments/ab/libsage355_$v.so
  rm -rf /tmp/tv_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tv_$v -- python3 $GRAFT_REPO_ROOT/experiments/mb_so.py > /tmp/tv_$v.log 2>&1 || { tail -5 /tmp/tv_$v.log; exit 1; }
  echo "$v: $(grep 'single stream' /tmp/tv_$v.log | sed 's/.*stream: //') | $(python3 $GRAFT_REPO_ROOT/experiments/trace_kernels.py /tmp/tv_$v 500)"
done
