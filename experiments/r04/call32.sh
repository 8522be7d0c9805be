#!/bin/bash
# the driver's command, 10 times on one box: the distribution of the recorded number on the final library
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c32
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04c32/b$i.json 2> gpurun_out/r04c32/b$i.err || { echo "run $i failed"; tail -3 gpurun_out/r04c32/b$i.err; exit 1; }
  echo "run $i done"
done
python3 - <<'PY'
import json, statistics as st
rows=[json.load(open(f'gpurun_out/r04c32/b{i}.json')) for i in range(1,11)]
a=[1e3*r['ms_per_step'] for r in rows]; c=[1e3*r['config']['variants']['configs3_rmat23']['ms_per_step'] for r in rows]
print('configs[2] us/step:', ' '.join('%.2f'%x for x in a), '-> mean %.2f sd %.2f'%(st.mean(a), st.stdev(a)))
print('configs[3] us/step:', ' '.join('%.2f'%x for x in c), '-> mean %.2f sd %.2f'%(st.mean(c), st.stdev(c)))
print('forward_frac: %.3f / %.3f' % (st.mean(r['roofline']['forward_frac'] for r in rows), st.mean(r['config']['variants']['configs3_rmat23']['forward_frac'] for r in rows)))
print('checks:', all(r['timed_path_check']['bit_identical_to_oracle_gated_forward'] and r['config']['variants']['configs3_rmat23']['timed_path_check']['bit_identical_to_oracle_gated_forward'] for r in rows))
PY
