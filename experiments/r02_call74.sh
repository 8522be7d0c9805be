#!/bin/bash
# bytes past L2 per forward at configs 4 and 5 (and 3 concat): does the pipeline's period follow them there too?
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD; O=$R/gpurun_out/law; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for cfg in "4" "5" "5 --mode concat" "3 --mode concat"; do
  tag=$(echo $cfg | tr -d ' -' )
  for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    name=${tag}_$(echo $group | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $group --output-format csv -d $O/$name -- python3 $R/bench.py --config $cfg --steps 12 --warmup 3 --exec direct --streams 1 --no-parity --cpu-seconds 0 --preheat-seconds 0 > $O/$name.log 2>&1 || { echo "$name failed"; tail -3 $O/$name.log; exit 1; }
  done
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/${tag}_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "gather" if "gather_mean" in n else "dense" if "dense_" in n else "layer2" if ("tile16" in n or "layer_fused" in n) else ("sample_outer" if "true, true>" in n else "sample_inner") if "sample_kernel" in n else None
        if k and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"): tot[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
s = 0; parts = []
for k, c in tot.items():
    rd = sum(c["FETCH_SIZE"][len(c["FETCH_SIZE"])//4:]) / max(1, len(c["FETCH_SIZE"][len(c["FETCH_SIZE"])//4:])) * 2048 / 1e6
    wr = sum(c["WRITE_SIZE"][len(c["WRITE_SIZE"])//4:]) / max(1, len(c["WRITE_SIZE"][len(c["WRITE_SIZE"])//4:])) * 1024 / 1e6
    s += rd + wr; parts.append("%s %.0f+%.0f" % (k, rd, wr))
print("config $cfg: %.0f MB past L2 per forward (%s)" % (s, ", ".join(parts)))
PY
done
