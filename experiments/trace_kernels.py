"""Median / mean duration of each libsage355 kernel in a rocprofv3 --kernel-trace output directory (last N launches)."""
import csv, glob, statistics, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 500
rows = [r for r in csv.DictReader(open(f)) if any(s in r["Kernel_Name"] for s in ("sample_kernel", "layer_fused", "layer_tile16", "gather_mean", "dense_layer", "dense_bf16x3"))]
rows = rows[-last:]
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    key = ("dense" if "dense_" in n else "tile16" if "tile16" in n else "fused" if "layer_fused" in n else "gather" if "gather" in n
           else "sample_outer" if "true, true>" in n else "sample_inner")
    by[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print("  ".join("%s %.1f/%.1f" % (k, statistics.median(v) / 1e3, sum(v) / len(v) / 1e3) for k, v in sorted(by.items())), "(median/mean us)")
