#!/bin/bash
# run-to-run spread of the two forms of the bench on ONE box: 16 runs each of the driver's 20-step form and of the 200-step default
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c53
for i in $(seq 16); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > gpurun_out/r03c53/a$i.json 2>/dev/null
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-variant --no-parity > gpurun_out/r03c53/b$i.json 2>/dev/null
done
python3 - <<'PY'
import json, glob, statistics as st
for tag, name in (("a", "20 steps after 5"), ("b", "200 steps after 20")):
    v = [json.load(open(f))["ms_per_step"] * 1e3 for f in sorted(glob.glob(f"gpurun_out/r03c53/{tag}*.json"))]
    print(f"{name}: n = {len(v)}, mean {st.mean(v):.2f} us, stdev {st.pstdev(v):.2f}, min {min(v):.2f}, max {max(v):.2f}")
PY
