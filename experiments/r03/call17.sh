#!/bin/bash
# r03 call 17: the driver's form (20 steps after 5), five interleaved repetitions: depth 4 / 5 / 6, per-batch submit vs one C loop
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c17; mkdir -p $O
for rep in 1 2 3 4 5; do for v in "SAGE_DEPTH=4" "SAGE_DEPTH=5" "SAGE_DEPTH=6" "SAGE_DEPTH=4 SAGE_SUBMIT_MANY=1" "SAGE_DEPTH=6 SAGE_SUBMIT_MANY=1"; do
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > $O/d.json 2> $O/d.err || tail -3 $O/d.err
  python3 -c "
import json; d=json.load(open('$O/d.json')); print('%-36s rep $rep: %.2f us/fwd host %.1f' % ('$v', 1e3*d['ms_per_step'], 1e3*d['config']['host_enqueue_ms_per_step']))"
done; done | tee $O/log.txt
python3 - <<'PY'
import collections, re
d = collections.defaultdict(list)
for ln in open("gpurun_out/r03c17/log.txt"):
    m = re.match(r"(.*?)\s+rep \d+: ([\d.]+) us", ln)
    if m: d[m.group(1).strip()].append(float(m.group(2)))
for k, v in d.items(): print("%-36s mean %.2f min %.2f max %.2f" % (k, sum(v) / len(v), min(v), max(v)))
PY
