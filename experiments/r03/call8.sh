#!/bin/bash
# r03 call 8: reproducible training tests; kernel breakdown of a training step at config-3 size
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c8; mkdir -p $O; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_engine_train.py tests/test_gpu_backward.py -x -q -m gpu -s -k "not config3_size_against and not reference_f1 and not drop_in" > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout -k 10 300 python experiments/train_big.py > $O/train_big.log 2>&1; grep -E "ms per|captured" $O/train_big.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/t -- python3 $GRAFT_REPO_ROOT/experiments/train_prof.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/$O/prof.log; exit 1; }
cd $GRAFT_REPO_ROOT; f=$(ls $O/t/*/*kernel_stats.csv | head -1); python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.1f us" % (tot / 40 / 1e3))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%8.1f us/step  calls/step %5.1f  avg %8.2f us  %s" % (float(r["TotalDurationNs"]) / 40 / 1e3, int(r["Calls"]) / 40, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
