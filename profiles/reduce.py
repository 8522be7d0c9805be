"""gpurun_out/<tag>/ (written by profiles/collect.sh) -> the summaries kept under profiles/.

  <tag>_bench.json, <tag>_bench_under_rocprof.json   the bench lines
  <tag>_kernel_stats.csv                             rocprofv3 --kernel-trace --stats summary (libsage355 kernels first)
  <tag>_pmc_per_kernel.json                          per-launch averages of the PMC counters, per kernel
  traffic.json                                       HBM-side bytes per launch of the dominant kernel (bench.py reads it)
Counter arithmetic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 16-B/lane loads at half size, so reads = FETCH_SIZE x 1024 x 2 (cross-checked here against
TCC_EA0_RDREQ_sum x 128 B, which must agree)."""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
OURS = ("sample_kernel", "gather_mean", "dense_bf16x3_kernel", "dense_layer_kernel", "layer_tile16_kernel", "layer_fused_kernel", "linear_act_kernel")


def short(name):
    for k in OURS:
        if k in name:
            if k == "sample_kernel":
                return "sample_kernel (outer hop)" if "true, true>" in name else "sample_kernel (inner hop)"
            if k == "gather_mean":
                return "gather_mean_sliced_kernel" if "sliced" in name else "gather_mean_kernel"
            return k
    return None


for f in ("bench.json", "bench_under_rocprof.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(here, f"{tag}_{f}"))

stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True))
if stats:
    rows = list(csv.DictReader(open(stats[-1])))
    rows.sort(key=lambda r: (short(r["Name"]) is None, -float(r["TotalDurationNs"])))
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:40])

per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "pmc_*", "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in per.items():
    d = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
    d["launches_averaged"] = min(len(v) for v in cs.values())
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1), 4)
    if "TCC_EA0_RDREQ_sum" in d:
        d["beyond_L2_read_bytes (TCC_EA0_RDREQ x 128)"] = d["TCC_EA0_RDREQ_sum"] * 128
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2, gfx950 half-count)"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        d["write_bytes (WRITE_SIZE KiB x 1024)"] = d["WRITE_SIZE"] * 1024
    out[k] = d
if out:
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_per_kernel.json"), "w"), indent=1)
    bench = json.load(open(os.path.join(here, f"{tag}_bench.json")))
    dom = bench["roofline"]["kernel"].split(" ")[0]
    d = out.get(dom)
    if d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        t = {"layer1_hbm_bytes_per_launch": int(d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024),
             "kernel": dom, "workload": bench["config"]["workload"],
             "read_bytes": int(d["FETCH_SIZE"] * 2048), "write_bytes": int(d["WRITE_SIZE"] * 1024),
             "l2_hit_rate": d.get("l2_hit_rate"), "source": f"profiles/{tag}_pmc_per_kernel.json",
             "method": "rocprofv3 --pmc, separate passes (profiles/collect.sh); FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 16-B/lane "
                       "loads at half size; TCC_EA0_RDREQ_sum x 128 B agrees) + WRITE_SIZE KiB x 1024",
             "note": "fabric-side counters: Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
        json.dump(t, open(os.path.join(here, "traffic.json"), "w"), indent=1)
        print("traffic:", t)
print("reduced", src, "->", here)
