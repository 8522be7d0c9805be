"""gpurun_out/<tag>/ (written by profiles/collect.sh) -> the summaries kept under profiles/.

  <tag>_bench*.json                                  the bench lines (default, 20/5 as the driver runs it, variants)
  <tag>_kernel_stats.csv                             rocprofv3 --kernel-trace --stats summary (libsage355 kernels first)
  <tag>_pmc_per_kernel.json                          per-launch averages of the PMC counters, per kernel, with derived figures
  traffic.json                                       HBM-side bytes per launch of the dominant kernel (bench.py reads it)
  mfma.json                                          matrix-pipe utilisation of the MFMA kernels (bench.py reads it)
Counter arithmetic follows /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 16-B/lane loads at half size, so reads = FETCH_SIZE x 1024 x 2 (cross-checked here against
TCC_EA0_RDREQ_sum x 128 B, which must agree).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the SIMDs
(= 32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32, 32 per v_mfma_f32_16x16x4_f32); SQ_INSTS_VALU_MFMA_MOPS_* x 512
= flops issued on the matrix pipe.  Busy fraction = busy cycles / (1024 SIMDs x kernel duration x clock), with the clock taken
as GRBM_GUI_ACTIVE / 8 XCDs / duration when that counter is in the pass (the guide warns that it reads high on dispatches
shorter than 0.3 ms, which makes the busy fraction a LOWER bound), else 2.4 GHz."""
import collections, csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
OURS = ("sample_fused_kernel", "sample_kernel", "gather_mean", "dense_bf16x3_kernel", "dense_pc_kernel", "dense_layer_kernel", "layer_tile16_kernel", "layer_fused_kernel",
        "linear_act_kernel", "prepare_weights_kernel")
PEAK_F32_TF, PEAK_BF16_TF, SIMDS = 157.3, 2500.0, 1024


def short(name):
    for k in OURS:
        if k in name:
            if k == "sample_fused_kernel":
                return "sample_fused_kernel (outer + inner hop)"
            if k == "sample_kernel":
                return "sample_kernel (outer hop)" if "true, true>" in name else "sample_kernel (inner hop)"
            if k == "gather_mean":
                return "gather_mean_rows_kernel" if "rows" in name else ("gather_mean_sliced_pipe_kernel" if "pipe" in name else
                                                                          ("gather_mean_sliced_kernel" if "sliced" in name else "gather_mean_kernel"))
            return k
    return None


for f in ("bench.json", "bench_under_rocprof.json", "bench_20_5.json", "bench_layout_input.json", "bench_replay2.json"):
    p = os.path.join(src, f)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(here, f"{tag}_{f}"))

# gpurun merges a call's outputs INTO gpurun_out/ (nothing is deleted there): of several runs' files take the newest one only
stats = sorted(glob.glob(os.path.join(src, "stats", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    rows = list(csv.DictReader(open(stats[-1])))
    rows.sort(key=lambda r: (short(r["Name"]) is None, -float(r["TotalDurationNs"])))
    for r in rows:
        if len(r["Name"]) > 160:
            r["Name"] = r["Name"][:157] + "..."
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:40])


stats1 = sorted(glob.glob(os.path.join(src, "stats_alone", "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats1:
    rows = list(csv.DictReader(open(stats1[-1])))
    rows.sort(key=lambda r: (short(r["Name"]) is None, -float(r["TotalDurationNs"])))
    for r in rows:
        if len(r["Name"]) > 160:
            r["Name"] = r["Name"][:157] + "..."
    with open(os.path.join(here, f"{tag}_kernel_stats_one_batch_in_flight.csv"), "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows[:40])


def per_kernel(pattern):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    by_dir = {}
    for f in glob.glob(os.path.join(src, pattern, "**", "*_counter_collection.csv"), recursive=True):
        d_ = os.path.dirname(f)                      # one pass = one directory: the newest file of each (older ones are earlier collections)
        if d_ not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[d_]):
            by_dir[d_] = f
    for f in by_dir.values():
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = {}
    for k, cs in per.items():
        d = {c: round(sum(v) / len(v), 1) for c, v in cs.items()}
        d["launches_averaged"] = min(len(v) for v in cs.values())
        d["duration_us_under_pmc"] = round(sum(dur[k]) / len(dur[k]) / 1e3, 2)
        out[k] = d
    return out


def derive(d):
    if "TCC_HIT_sum" in d and "TCC_MISS_sum" in d:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / max(d["TCC_HIT_sum"] + d["TCC_MISS_sum"], 1), 4)
    if "TCC_EA0_RDREQ_sum" in d:
        d["beyond_L2_read_bytes (TCC_EA0_RDREQ x 128)"] = d["TCC_EA0_RDREQ_sum"] * 128
    if "FETCH_SIZE" in d:
        d["hbm_read_bytes (FETCH_SIZE KiB x 1024 x 2, gfx950 half-count)"] = d["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in d:
        d["write_bytes (WRITE_SIZE KiB x 1024)"] = d["WRITE_SIZE"] * 1024
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0:
        dur_s = d["duration_us_under_pmc"] * 1e-6
        clock = d["GRBM_GUI_ACTIVE"] / 8 / dur_s if d.get("GRBM_GUI_ACTIVE") else 2.4e9
        m = {"mfma_busy_cycles_per_launch": d["SQ_VALU_MFMA_BUSY_CYCLES"],
             "busy_fraction_of_all_simds": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * dur_s * clock), 4),
             "clock_GHz_used (GRBM_GUI_ACTIVE / 8 / duration; reads high on short dispatches)": round(clock / 1e9, 2),
             "busy_fraction_at_2.4GHz": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * dur_s * 2.4e9), 4)}
        bf, f32 = d.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0), d.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        if bf or f32:
            m["issued_TFLOPs"] = round((bf + f32) * 512 / dur_s / 1e12, 1)
            m["issued_frac_of_peak"] = round((bf * 512 / PEAK_BF16_TF + f32 * 512 / PEAK_F32_TF) / dur_s / 1e12, 4)
            if bf:      # split-bf16: six bf16 MFMAs per fp32-equivalent product
                m["fp32_equivalent_TFLOPs"] = round(bf * 512 / 6 / dur_s / 1e12, 1)
                m["fp32_equivalent_frac_of_157TF"] = round(bf * 512 / 6 / dur_s / 1e12 / PEAK_F32_TF, 4)
        d["mfma"] = m
    return d


out = {k: derive(d) for k, d in per_kernel("pmc_*").items()}
bench = None
bp = os.path.join(here, f"{tag}_bench.json")
if os.path.exists(bp):
    try:
        bench = json.load(open(bp))
    except Exception:
        bench = None
if out:
    json.dump(out, open(os.path.join(here, f"{tag}_pmc_per_kernel.json"), "w"), indent=1)
    dom = next((k for k in ("gather_mean_sliced_pipe_kernel", "gather_mean_rows_kernel", "gather_mean_sliced_kernel") if k in out), None)
    d = out.get(dom)
    if bench and d and "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        t = {"layer1_hbm_bytes_per_launch": int(d["FETCH_SIZE"] * 2048 + d["WRITE_SIZE"] * 1024),
             "kernel": dom, "workload": bench["config"]["workload"], "engine_layout": bench["config"]["engine_layout"].split(" ")[0],
             "read_bytes": int(d["FETCH_SIZE"] * 2048), "write_bytes": int(d["WRITE_SIZE"] * 1024),
             "l2_hit_rate": d.get("l2_hit_rate"), "source": f"profiles/{tag}_pmc_per_kernel.json",
             "method": "rocprofv3 --pmc, separate passes (profiles/collect.sh); FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 16-B/lane "
                       "loads at half size; TCC_EA0_RDREQ_sum x 128 B agrees) + WRITE_SIZE KiB x 1024",
             "note": "fabric-side counters: Infinity-Cache hits are included, so this is an upper bound on HBM bytes"}
        json.dump(t, open(os.path.join(here, "traffic.json"), "w"), indent=1)
        print("traffic:", t)
    mf = {k: v["mfma"] | {"duration_us_under_pmc": v["duration_us_under_pmc"]} for k, v in out.items() if "mfma" in v}
    c5 = {k: derive(d) for k, d in per_kernel("c5_pmc_*").items()}
    if c5:
        json.dump(c5, open(os.path.join(here, f"{tag}_pmc_per_kernel_config5.json"), "w"), indent=1)
    if bench and mf:
        mj = {"workload": bench["config"]["workload"], "kernels": mf,
              "config5_contraction_in_isolation (D0 = 100, H = 128, fanout 20/25)":
                  {k: v["mfma"] | {"duration_us_under_pmc": v["duration_us_under_pmc"]} for k, v in c5.items() if "mfma" in v},
              "peaks": {"fp32_mfma_TF": PEAK_F32_TF, "bf16_mfma_dense_TF": PEAK_BF16_TF, "simds": SIMDS},
              "source": f"profiles/{tag}_pmc_per_kernel.json, profiles/{tag}_pmc_per_kernel_config5.json (rocprofv3 --pmc, one stream)"}
        json.dump(mj, open(os.path.join(here, "mfma.json"), "w"), indent=1)
        print("mfma:", json.dumps(mf)[:600])
print("reduced", src, "->", here)
