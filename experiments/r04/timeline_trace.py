"""Per-batch timeline of the LAST phase of a rocprofv3 --kernel-trace of bench.py that holds exactly N forwards (the driver's 20-step timed region):
start / end of every kernel relative to the first kernel of the region, and the interval between consecutive layer-2 ends.
    python experiments/r04/timeline_trace.py <trace dir> <N>"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
N = int(sys.argv[2])
def kind(n):
    if "sample_kernel" in n: return "So" if "true, true>" in n else "Si"
    if "gather_mean" in n: return "G"
    if "dense_" in n: return "D"
    if "layer_tile16" in n or "layer_fused" in n: return "L"
    return None
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"])) for r in csv.DictReader(open(f)) if kind(r["Kernel_Name"]))
phases = [[rows[0]]]; last = rows[0][1]
for r in rows[1:]:
    if r[0] - last > 150_000: phases.append([])
    phases[-1].append(r); last = max(last, r[1])
print("phases (kernels):", " ".join(str(len(p)) for p in phases if len(p) >= 25))
cands = [p for p in phases if len(p) == 5 * N]
for p in cands[:2]:
    by = collections.defaultdict(list)
    for s, e, k in p: by[k].append((s, e))
    t0 = p[0][0]
    span = (max(e for _, e, _ in p) - t0) / 1e3
    print(f"region of {N} forwards: {span:.0f} us = {span / N:.1f} us per step (under the tracer)")
    prev = 0.0
    for b in range(N):
        t = {k: ((by[k][b][0] - t0) / 1e3, (by[k][b][1] - t0) / 1e3) for k in by}
        print(f"  b={b:2d} " + "  ".join(f"{k} {t[k][0]:6.0f}-{t[k][1]:6.0f}" for k in ("So", "Si", "G", "D", "L")) + f"   L-end delta {t['L'][1] - prev:6.1f}")
        prev = t["L"][1]
