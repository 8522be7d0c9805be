"""Timeline of the timed region of `bench.py --steps 20 --warmup 5` from a rocprofv3 --kernel-trace run: where do the ~5 us per step
between the 20-step form and the steady state go (fill, drain, gaps on the gather's queue)?   python experiments/r03/trace20.py <dir> [steps]"""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
def kind(n):
    if "sample_kernel" in n: return "So" if "true, true>" in n else "Si"
    if "gather_mean" in n: return "G"
    if "dense_" in n: return "D"
    if "layer_tile16" in n or "layer_fused" in n: return "L2"
    return None
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind(r["Kernel_Name"]), r["Queue_Id"]) for r in csv.DictReader(open(f)) if kind(r["Kernel_Name"]))
phases = [[rows[0]]]; last = rows[0][1]
for r in rows[1:]:
    if r[0] - last > 100_000: phases.append([])
    phases[-1].append(r); last = max(last, r[1])
counts = [sum(1 for r in p if r[2] == "G") for p in phases]
print(len(phases), "phases; gathers per phase:", counts)
cands = [p for p in phases if sum(1 for r in p if r[2] == "G") == steps]
if not cands:          # warm-up and timed region in one phase (the fence between them is short): cut at the largest gap on the gather queue
    for p in phases:
        gs = [r for r in p if r[2] == "G"]
        if len(gs) > steps:
            gaps = [(b[0] - a[1], i) for i, (a, b) in enumerate(zip(gs, gs[1:]))]
            cut = gs[max(gaps[len(gs) - steps - 1:len(gs) - steps])[1] + 1][0] if len(gs) - steps - 1 >= 0 else gs[0][0]
            # everything from the first sampler kernel that precedes the first timed gather
            first_g = gs[len(gs) - steps]
            start = max(r[0] for r in p if r[2] == "So" and r[0] < first_g[0])
            cands.append([r for r in p if r[0] >= start])
for p in cands[:3]:
    t0 = p[0][0]
    span = (max(r[1] for r in p) - t0) / 1e3
    print(f"phase span {span:.1f} us = {span / steps:.2f} us per step (kernel start of the first sampler -> end of the last layer 2)")
    gs = [r for r in p if r[2] == "G"]
    print("  first gather starts at %.1f us; last gather ends at %.1f us; tail after it %.1f us" % ((gs[0][0] - t0) / 1e3, (gs[-1][1] - t0) / 1e3, span - (gs[-1][1] - t0) / 1e3))
    print("  gather durations:", " ".join("%.0f" % ((e - s) / 1e3) for s, e, _, _ in gs))
    print("  gaps between gathers:", " ".join("%.1f" % ((b[0] - a[1]) / 1e3) for a, b in zip(gs, gs[1:])))
    for k in ("So", "Si", "D", "L2"):
        ks = [r for r in p if r[2] == k]
        print(f"  {k:3s} durations:", " ".join("%.0f" % ((e - s) / 1e3) for s, e, _, _ in ks))
    for s, e, k, q in p[:24] + p[-12:]:
        print("     %8.1f -> %8.1f  (%5.1f us) %-3s q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, q))
