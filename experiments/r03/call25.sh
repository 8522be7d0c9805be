#!/bin/bash
# r03 call 25: in-pipeline and alone kernel stats with the slice-major default; PMC of the new gather
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out/r03c25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --cpu-seconds 0 --no-variant > $O/b.json 2> $O/stats.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/alone -- python3 $R/bench.py --exec direct --streams 1 --steps 100 --warmup 10 --no-parity --cpu-seconds 0 --no-variant > $O/a.json 2> $O/alone.log
PMC="python3 $R/bench.py --steps 12 --warmup 3 --exec direct --streams 1 --no-parity --cpu-seconds 0 --preheat-seconds 0 --no-variant"
for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  name=pmc_$(echo $group | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $O/$name -- $PMC > $O/$name.log 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, json, collections
O = "gpurun_out/r03c25"
for d in ("stats", "alone"):
    f = sorted(glob.glob(f"{O}/{d}/**/*_kernel_stats.csv", recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in ("sample_kernel", "gather_mean", "dense_", "layer_tile16"))]
    print(d, [(r["Name"].split("::")[-1][:26], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), round(float(r["MinNs"]) / 1e3, 1)) for r in rows])
print(json.load(open(f"{O}/b.json"))["ms_per_step"])
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{O}/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gather_mean" in r["Kernel_Name"]: per["gather"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in per.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    print(k, {c: round(v, 1) for c, v in d.items()}, "read MB", d.get("FETCH_SIZE", 0) * 2048 / 1e6, "write MB", d.get("WRITE_SIZE", 0) * 1024 / 1e6, "hit", d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1))
PY
find $O -name "*_kernel_trace.csv" -size +4M -delete; find $O -name "*_counter_collection.csv" -size +4M -delete
