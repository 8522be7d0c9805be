#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c19; mkdir -p $O
for rep in 1 2 3 4 5 6 7 8; do for v in "SAGE_DEPTH=4" "SAGE_DEPTH=8"; do
  env $v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > $O/d.json 2> $O/d.err || tail -3 $O/d.err
  python3 -c "
import json; d=json.load(open('$O/d.json')); print('%-14s rep $rep: %.2f' % ('$v', 1e3*d['ms_per_step']))"
done; done | tee $O/log.txt
python3 - <<'PY'
import collections, re
d = collections.defaultdict(list)
for ln in open("gpurun_out/r03c19/log.txt"):
    m = re.match(r"(\S+)\s+rep \d+: ([\d.]+)", ln)
    if m: d[m.group(1)].append(float(m.group(2)))
for k, v in sorted(d.items()): print(k, "mean %.2f min %.2f max %.2f" % (sum(v) / len(v), min(v), max(v)), sorted(v))
PY
