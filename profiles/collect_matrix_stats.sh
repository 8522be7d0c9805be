#!/bin/bash
# Per-kernel rocprofv3 summaries of the OTHER configurations, in the running pipeline and with one batch in flight (VERDICT r2 #8:
# every r0N_matrix_* line's dominant-kernel fraction must be recomputable):  gpurun -- 'bash profiles/collect_matrix_stats.sh r03'
# -> gpurun_out/<tag>_mstats/<name>[_alone]/..., reduced into profiles/<tag>_kernel_stats_<name>[_one_batch_in_flight].csv
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_mstats; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run(){ name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$R/bench.py" --steps 100 --cpu-seconds 0 --no-variant "$@" \
      > "$O/$name.json" 2> "$O/$name.log"; echo "$name rc=$?"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_alone" -- python3 "$R/bench.py" --steps 60 --warmup 10 --exec direct --streams 1 \
      --no-parity --cpu-seconds 0 --no-variant "$@" > "$O/${name}_alone.json" 2> "$O/${name}_alone.log"; echo "${name}_alone rc=$?"
}
run c3_concat --config 3 --mode concat
run c4_rmat23 --config 4
run c5_gcn --config 5
run c5_concat --config 5 --mode concat
python3 - "$O" "$TAG" "$R/profiles" <<'PY'
import csv, glob, os, sys
src, tag, here = sys.argv[1:4]
OURS = ("sample_", "gather_mean", "dense_bf16x3", "dense_layer", "layer_tile16", "layer_fused", "linear_act", "prepare_weights")
for d in sorted(os.listdir(src)):
    p = os.path.join(src, d)
    if not os.path.isdir(p):
        continue
    files = sorted(glob.glob(os.path.join(p, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[-1])))
    rows.sort(key=lambda r: (not any(k in r["Name"] for k in OURS), -float(r["TotalDurationNs"])))
    for r in rows:
        if len(r["Name"]) > 160:
            r["Name"] = r["Name"][:157] + "..."
    name = d[:-6] + "_one_batch_in_flight" if d.endswith("_alone") else d
    with open(os.path.join(here, f"{tag}_kernel_stats_{name}.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:24])
    ours = [r for r in rows if any(k in r["Name"] for k in OURS)][:6]
    print(name, [(r["Name"][:40], round(float(r["AverageNs"]) / 1e3, 1)) for r in ours])
PY
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
