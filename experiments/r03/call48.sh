#!/bin/bash
# final checks: full GPU test suite, smoke(), the bench lines again (roofline.traffic now from the regenerated profiles/traffic.json)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r03_final_tests.log 2>&1; tail -4 gpurun_out/r03_final_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -4
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err; echo "bench20 rc=$?"
timeout -k 10 300 python bench.py --cpu-seconds 0 --engine-layout input > $O/bench_layout_input.json 2> $O/bench_layout_input.err
timeout -k 10 300 python bench.py --cpu-seconds 0 --exec replay --streams 2 > $O/bench_replay2.json 2> $O/bench_replay2.err
python3 - <<'PY'
import json
for f in ("bench","bench_20_5","bench_layout_input","bench_replay2"):
    d=json.load(open(f"gpurun_out/r03/{f}.json")); r=d["roofline"]
    print(f, round(d["ms_per_step"]*1e3,2), "us", "%.3g"%d["value"], "fwd_frac", r["forward_frac"], "frac", r["frac"], "alone", r["frac_alone"], "traffic", r["traffic"], "cpu", (d.get("cpu_baseline") or {}).get("value"))
PY
