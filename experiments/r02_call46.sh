#!/bin/bash
# what does one submit cost the HOST?  short runs (no queue ever fills) of: everything, S+L only, events only
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/ab
for st in 20 60 400; do STEPS=$st bash experiments/ab_run.sh 1 base skipGD skipALL > gpurun_out/ab/h$st.txt; python3 - <<PY
import json
for n in ("base","skipGD","skipALL"):
    d=json.load(open("gpurun_out/ab/%s.1.json" % n)); print("steps $st %-8s step %.1f us, host enqueue %.1f us" % (n, d["ms_per_step"]*1e3, d["config"]["host_enqueue_ms_per_step"]*1e3))
PY
done
