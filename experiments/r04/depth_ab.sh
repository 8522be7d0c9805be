#!/bin/bash
# pipeline depth on the final library (express lane + tail events), both forms, interleaved
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c51
run() { name=$1; shift
  for form in short long; do
    if [ $form = long ]; then a="--steps 300 --warmup 50"; else a="--steps 20 --warmup 5"; fi
    env "$@" timeout -k 10 300 python bench.py $a --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c51/$name.$form.$rep.json 2> gpurun_out/r04c51/$name.$form.$rep.err || { echo "$name FAILED"; tail -3 gpurun_out/r04c51/$name.$form.$rep.err; return 0; }
  done
  python3 -c "
import json
l=json.load(open('gpurun_out/r04c51/$name.long.$rep.json')); s=json.load(open('gpurun_out/r04c51/$name.short.$rep.json'))
print('rep $rep %-8s 20-step %6.2f  300-step %6.2f' % ('$name', 1e3*s['ms_per_step'], 1e3*l['ms_per_step']))"
}
for rep in 1 2 3; do
  run depth4 SAGE_DEPTH=4
  run depth3 SAGE_DEPTH=3
  run depth5 SAGE_DEPTH=5
  run depth6 SAGE_DEPTH=6
  run depth8 SAGE_DEPTH=8
done 2>&1 | tee gpurun_out/r04c51/ab.log
python3 - <<'PY'
import re, statistics as st, collections
d=collections.defaultdict(lambda: ([],[]))
for ln in open('gpurun_out/r04c51/ab.log'):
    m=re.match(r'rep \d (\S+)\s+20-step\s+([\d.]+)\s+300-step\s+([\d.]+)', ln)
    if m: d[m.group(1)][0].append(float(m.group(2))); d[m.group(1)][1].append(float(m.group(3)))
for k,(a,b) in d.items(): print('%-8s 20-step mean %.2f  300-step mean %.2f' % (k, st.mean(a), st.mean(b)))
PY
