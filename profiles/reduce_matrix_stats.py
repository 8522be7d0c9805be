"""gpurun_out/<tag>_mstats/ (written by profiles/collect_matrix_stats.sh) -> profiles/<tag>_kernel_stats_<cfg>[_one_batch_in_flight].csv
    python profiles/reduce_matrix_stats.py gpurun_out/r03_mstats r03"""
import csv, glob, os, sys
src, tag = sys.argv[1:3]
here = os.path.dirname(os.path.abspath(__file__))
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
    print(name, [(r["Name"].split("::")[-1][:28], round(float(r["AverageNs"]) / 1e3, 1)) for r in ours])
