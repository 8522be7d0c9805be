"""avg of every PMC counter for the gather kernel from a rocprofv3 --pmc csv dir:  python pmc_gather.py <dir> [kernel substring]"""
import csv, glob, sys, collections
sub = sys.argv[2] if len(sys.argv) > 2 else "gather_mean"
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
d = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in acc.items()}
out = dict(d)
if "TCC_HIT_sum" in d: out["l2_hit"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
if "TCC_EA0_RDREQ_sum" in d: out["beyond_L2_read_MB"] = round(d["TCC_EA0_RDREQ_sum"] * 128 / 1e6, 1)
if "FETCH_SIZE" in d: out["fetch_MB(x2)"] = round(d["FETCH_SIZE"] * 2048 / 1e6, 1)
if "WRITE_SIZE" in d: out["write_MB"] = round(d["WRITE_SIZE"] * 1024 / 1e6, 1)
print({k: (round(v, 1) if isinstance(v, float) else v) for k, v in out.items()})
