"""Who waits for whom in the running role pipeline?  Reads a rocprofv3 --kernel-trace CSV of bench.py, numbers the kernels of every kind
in order (= batch index), and for every kernel start reports which of its constraints was the LAST to be met -- the previous kernel on its
own stream, its producer (S -> G -> D -> L) or the workspace release (L(b - depth) -> So(b)) -- and how long after that it started.
    python experiments/r04/dep_trace.py <trace dir> [depth] [skip_first_n_batches]"""
import csv, glob, sys, collections
import numpy as np
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 4
def kind(n):
    if "sample_kernel" in n: return "So" if "true, true>" in n else "Si"
    if "gather_mean" in n: return "G"
    if "dense_" in n: return "D"
    if "layer_tile16" in n or "layer_fused" in n: return "L"
    return None
rows = []
for r in csv.DictReader(open(f)):
    k = kind(r["Kernel_Name"])
    if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k))
rows.sort()
# the last contiguous phase (the timed region follows the warm-up without a long gap; keep the last 60 %)
phases = [[rows[0]]]; last = rows[0][1]
for r in rows[1:]:
    if r[0] - last > 300_000: phases.append([])
    phases[-1].append(r); last = max(last, r[1])
# the timed region + its warm-up: the LAST phase that holds 100 .. 1000 forwards (the preheat phases hold thousands, the profiled passes follow it)
want = int(sys.argv[4]) if len(sys.argv) > 4 else None
cands = [ph for ph in phases if 100 * 5 <= len(ph) <= 1000 * 5]
p = phases[want] if want is not None else (cands[0] if cands else max(phases, key=len))
print("phases (kernels): " + " ".join(str(len(ph)) for ph in phases if len(ph) >= 50))
by = collections.defaultdict(list)
for s, e, k in p: by[k].append((s, e))
n = min(len(v) for v in by.values())
assert all(len(v) == n for v in by.values()), {k: len(v) for k, v in by.items()}
skip = int(sys.argv[3]) if len(sys.argv) > 3 else n // 3
import os
with open(os.path.join(sys.argv[1], "timed_phase.csv"), "w") as fh:          # small enough to keep: the analysed phase only
    fh.write("start_ns,end_ns,kind\n")
    t00 = p[0][0]
    for s, e, k in p: fh.write(f"{s - t00},{e - t00},{k}\n")
print(f"{len(phases)} phases; analysed phase: {n} batches, statistics over batches {skip} .. {n - depth - 1}")
S = {k: np.array(by[k][:n], dtype=np.float64) / 1e3 for k in by}           # us
def stat(x): x = np.asarray(x); return "avg %5.1f med %5.1f p90 %5.1f" % (x.mean(), np.median(x), np.percentile(x, 90))
period = (S["L"][n - depth - 1, 1] - S["L"][skip, 1]) / (n - depth - 1 - skip)
print("period %.1f us" % period)
for k in ("So", "Si", "G", "D", "L"):
    d = S[k][skip:n - depth, 1] - S[k][skip:n - depth, 0]
    print(f"  {k:2s} duration {stat(d)}   busy {100 * d.mean() / period:4.0f} % of the period")
prod = {"Si": "So", "G": "Si", "D": "G", "L": "D"}
prev_same = {"So": "Si", "Si": "So", "G": "G", "D": "D", "L": "L"}          # stream S runs So(b) Si(b) So(b+1) ...
for k in ("So", "Si", "G", "D", "L"):
    waits, who = [], collections.Counter()
    for b in range(skip, n - depth):
        start = S[k][b, 0]
        cons = {}
        if k == "So":
            cons["stream (Si of the previous batch)"] = S["Si"][b - 1, 1]
            cons["workspace release (L of batch b - depth)"] = S["L"][b - depth, 1]
        elif k == "Si":
            cons["stream (So of this batch)"] = S["So"][b, 1]
        else:
            cons["stream (previous " + k + ")"] = S[k][b - 1, 1]
            cons["producer " + prod[k]] = S[prod[k]][b, 1]
        last_name = max(cons, key=cons.get)
        who[last_name] += 1
        waits.append(start - cons[last_name])
    print(f"  {k:2s} starts {stat(waits)} us after its last constraint; last constraint: " + ", ".join(f"{a}: {c}" for a, c in who.most_common()))
# what runs beside what: average number of kernels resident
ev = sorted([(s, 1) for s, e, _ in p] + [(e, -1) for s, e, _ in p])
run = 0; t_prev = ev[0][0]; area = 0
for t, d in ev:
    area += run * (t - t_prev); t_prev = t; run += d
print("  average kernels resident %.2f" % (area / (ev[-1][0] - ev[0][0])))
t0 = S["So"][skip, 0]
print("  timeline of three batches (us from So(b) start): ")
for b in range(skip, skip + 3):
    print("   b=%d " % b + "  ".join(f"{k} {S[k][b, 0] - t0:7.1f}->{S[k][b, 1] - t0:7.1f}" for k in ("So", "Si", "G", "D", "L")))
