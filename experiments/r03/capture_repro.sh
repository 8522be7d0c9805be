#!/bin/bash
# every capture pattern once, one process each; a crash of one does not stop the others (host-side faults only: trivial kernels)
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03_capture; mkdir -p $O
for p in ${PATTERNS:-0 1 2 3 4 5 6 7 8 9 10 11 12}; do
  echo "== pattern $p"
  timeout -k 5 60 experiments/r03/capture_repro $p > $O/p$p.log 2>&1; rc=$?
  cat $O/p$p.log; echo "   exit status $rc"
done 2>&1 | tee $O/summary.log
exit 0
