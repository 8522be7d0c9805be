"""Timeline of a rocprofv3 --kernel-trace run of pipe_sweep.py / bench.py: per kernel kind the duration, the queue it
ran on, idle gaps on its queue, and what it overlapped with.   python experiments/pipe_trace.py <trace dir> [skip_first_n]"""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
def kind(n):
    if "sample_kernel" in n or "sample_fused" in n: return "So" if ("true, true>" in n or "fused" in n) else "Si"
    if "gather_mean" in n or "gather_plus" in n: return "G"
    if "dense_" in n: return "D"
    if "layer_fused" in n or "layer_tile16" in n: return "L2"
    return None
rows = []
for r in csv.DictReader(open(f)):
    k = kind(r["Kernel_Name"])
    if k: rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k, r["Queue_Id"]))
rows.sort()
# split into phases at idle gaps > 300 us
phases = [[rows[0]]]; last = rows[0][1]
for r in rows[1:]:
    if r[0] - last > 300_000: phases.append([])
    phases[-1].append(r); last = max(last, r[1])
for p in phases:
    if len(p) < 200: continue
    q = p[len(p) // 4: -len(p) // 8]                       # steady part
    n_fwd = sum(1 for r in q if r[2] == "G")
    span = q[-1][1] - q[0][0]
    print(f"phase: {len(p)} kernels, steady part {len(q)} kernels / {n_fwd} gathers: {span / 1e3 / max(n_fwd, 1):.1f} us per forward")
    dur = collections.defaultdict(list); ov = collections.defaultdict(lambda: collections.defaultdict(float)); queues = collections.defaultdict(set)
    for i, (s, e, k, qu) in enumerate(q):
        dur[k].append(e - s); queues[k].add(qu)
        for j in range(max(0, i - 16), min(len(q), i + 16)):
            if j == i: continue
            s2, e2, k2, _ = q[j]
            o = min(e, e2) - max(s, s2)
            if o > 0: ov[k][k2] += o
    for k in ("So", "Si", "G", "D", "L2"):
        v = sorted(dur[k])
        if not v: continue
        tot = sum(v)
        print("  %3s n=%4d  avg %5.1f  min %5.1f  med %5.1f  max %5.1f us  queues %s | overlap: " % (k, len(v), tot / len(v) / 1e3, v[0] / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3, sorted(queues[k]))
              + "  ".join("%s %.2f" % (k2, o / tot) for k2, o in sorted(ov[k].items())))
    # idle time of the gather queue: gap between consecutive gathers
    gs = [r for r in q if r[2] == "G"]
    gaps = sorted((b[0] - a[1]) / 1e3 for a, b in zip(gs, gs[1:]))
    if gaps: print("  gap between consecutive gathers: avg %.1f  med %.1f  max %.1f us" % (sum(gaps) / len(gaps), gaps[len(gaps) // 2], gaps[-1]))
    # chip-level: fraction of time with >= 1 kernel running, and average number running
    ev = sorted([(s, 1) for s, e, _, _ in q] + [(e, -1) for s, e, _, _ in q])
    run = 0; t_prev = ev[0][0]; busy = 0; area = 0
    for t, d in ev:
        if run > 0: busy += t - t_prev
        area += run * (t - t_prev); t_prev = t; run += d
    print("  some kernel running %.1f %% of the time; average kernels running %.2f" % (100 * busy / span, area / span))
    # a sample of the timeline
    t0 = q[0][0]
    for s, e, k, qu in q[:30]:
        print("     %8.1f -> %8.1f  (%5.1f us) %-3s q%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, k, qu))
