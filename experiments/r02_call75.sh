#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out/trainprof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/experiments/train_prof.py > $O/log.txt 2>&1 || { tail -5 $O/log.txt; exit 1; }
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$O/t/**/*_kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per step: %.1f us" % (tot / 40 / 1e3))
for r in rows[:22]:
    print("%7.1f us/step  calls/step %5.1f  avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / 40 / 1e3, int(r["Calls"]) / 40, float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
rm -rf $O/t
