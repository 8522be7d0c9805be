"""Read a rocprofv3 kernel trace of a 2-stream bench run: per kernel type, the average duration inside the timed graph
replay phase and with which other kernel types its run time overlapped (fractions of its own duration)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
def kind(n):
    if "sample_kernel" in n: return "So" if "true, true>" in n else "Si"
    if "gather_mean" in n: return "G"
    if "dense_" in n: return "D"
    if "layer_fused" in n or "layer_tile16" in n: return "L2"
    return None
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"])) for r in csv.DictReader(open(f))]
rows = sorted(r for r in rows if r[2])
last = rows[0][1]; phases = [[rows[0]]]
for r in rows[1:]:
    if r[0] - last > 500_000: phases.append([])
    phases[-1].append(r); last = max(last, r[1])
for p in phases:
    if len(p) < 500: continue
    p = p[100:]
    dur = collections.defaultdict(list); ov = collections.defaultdict(lambda: collections.defaultdict(float))
    for i, (s, e, k) in enumerate(p):
        dur[k].append(e - s)
        for j in range(max(0, i - 12), min(len(p), i + 12)):
            if j == i: continue
            s2, e2, k2 = p[j]
            o = min(e, e2) - max(s, s2)
            if o > 0: ov[k][k2] += o
    span = p[-1][1] - p[0][0]
    print("phase of", len(p), "kernels: us/forward %.1f" % (span / 1e3 / (len(p) / 5)))
    for k in ("So", "Si", "G", "D", "L2"):
        v = dur[k]; tot = sum(v)
        print("  %3s n=%4d avg %.1f us | overlapped with: " % (k, len(v), tot / len(v) / 1e3) + "  ".join("%s %.2f" % (k2, o / tot) for k2, o in sorted(ov[k].items())))
