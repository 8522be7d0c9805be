#!/bin/bash
# the skip-stage experiment of round 2 again, now that the host is off the critical path (one enqueue thread per role stream):
# the pipeline without the contraction's kernel / without the gather's / without both (stale data downstream)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c37
STEPS=300 bash experiments/ab_run.sh 2 base skipD skipG skipGD 2>&1 | cut -c1-120 | tee gpurun_out/r03c37/ab.log
python3 - <<'PY'
import json,glob
for n in ("base","skipD","skipG","skipGD"):
    d=json.load(open(f"gpurun_out/ab/{n}.1.json")); print(n, "host enqueue us", 1e3*d["config"]["host_enqueue_ms_per_step"], "submit", 1e3*d["config"].get("host_submit_ms_per_step",0))
PY
