import csv, glob, statistics, collections, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("k_")]
rows = rows[-int(sys.argv[2]):] if len(sys.argv) > 2 else rows[-24 * 15:]
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    by[(r["Kernel_Name"], r["Grid_Size_X"], r["Workgroup_Size_X"], r["LDS_Block_Size"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
gaps = [int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]) for i in range(len(rows) - 1)]
for k, v in sorted(by.items()): print(k, "median %.2f us" % (statistics.median(v) / 1e3))
print("gap between consecutive kernels: median %.2f us" % (statistics.median(gaps) / 1e3))
