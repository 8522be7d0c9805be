"""Read a rocprofv3 kernel trace of a 2-stream bench run: per kernel type, the average duration and with which
other kernel types its run time overlapped (fractions of its own duration)."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
def kind(n):
    if "sample_kernel" in n: return "So" if "true, true>" in n or ", true, true" in n else "Si"
    if "gather_mean" in n: return "G"
    if "dense_layer" in n: return "D"
    if "layer_fused" in n: return "L2"
    return None
rows = []
for r in csv.DictReader(open(f)):
    k = kind(r["Kernel_Name"])
    if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
rows = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -1500:]
dur = collections.defaultdict(list); ov = collections.defaultdict(lambda: collections.defaultdict(float))
for i, (s, e, k) in enumerate(rows):
    dur[k].append(e - s)
    for j in range(max(0, i - 12), min(len(rows), i + 12)):
        if j == i: continue
        s2, e2, k2 = rows[j]
        o = min(e, e2) - max(s, s2)
        if o > 0: ov[k][k2] += o
span = rows[-1][1] - rows[0][0]
print("kernels", len(rows), "span us %.1f" % (span / 1e3), "-> us per forward %.1f" % (span / 1e3 / (len(rows) / 5)))
busy = 0; cur_e = rows[0][0]
for s, e, k in rows:
    if e > cur_e: busy += e - max(s, cur_e); cur_e = e
print("GPU busy fraction %.3f" % (busy / span))
for k, v in dur.items():
    tot = sum(v)
    print("%3s n=%4d avg %.1f us | overlapped with: " % (k, len(v), tot / len(v) / 1e3) + "  ".join("%s %.2f" % (k2, o / tot) for k2, o in sorted(ov[k].items())))
