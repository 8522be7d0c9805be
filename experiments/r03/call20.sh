#!/bin/bash
# r03 call 20: slice-major second copy of the engine's table for the column-sliced gather (SAGE_TABLE_SLICED=1): parity, then A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c20; mkdir -p $O
SAGE_TABLE_SLICED=1 timeout -k 10 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_forward.py -x -q -m gpu -k "config3_size_against or config3_shape or small_rmat" > $O/tests.log 2>&1; tail -4 $O/tests.log
grep -q "passed" $O/tests.log || exit 1
for rep in 1 2 3 4; do for v in "SAGE_TABLE_SLICED=0" "SAGE_TABLE_SLICED=1"; do for sw in "200 20" "20 5"; do set -- $sw
  env $v timeout -k 10 300 python bench.py --steps $1 --warmup $2 --cpu-seconds 0 --no-variant > $O/d.json 2> $O/d.err || { tail -3 $O/d.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/d.json')); r=d['roofline']; print('%-20s steps %3d rep $rep: %.2f us  G in situ %.1f alone %.1f  parity %.1e' % ('$v', $1, 1e3*d['ms_per_step'], 1e3*r['kernel_ms'], 1e3*r['kernel_ms_alone'], d['parity_max_err_vs_fp64_oracle']))"
done; done; done | tee $O/log.txt
python3 - <<'PY'
import collections, re
d = collections.defaultdict(list)
for ln in open("gpurun_out/r03c20/log.txt"):
    m = re.match(r"(\S+)\s+steps\s+(\d+) rep \d+: ([\d.]+) us  G in situ ([\d.]+) alone ([\d.]+)", ln)
    if m: d[(m.group(1), m.group(2))].append((float(m.group(3)), float(m.group(5))))
for k, v in sorted(d.items()): print(k, "period mean %.2f  (min %.2f max %.2f)  G alone %.1f" % (sum(x[0] for x in v) / len(v), min(x[0] for x in v), max(x[0] for x in v), sum(x[1] for x in v) / len(v)))
PY
