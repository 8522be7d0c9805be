#!/bin/bash
# (1) the N = 2 line as the driver would launch it, two ranks sharing this box's one GPU (gloo), configs[3] variant included;
# (2) run-to-run spread of both forms on this box, 12 runs each; (3) per-kernel stats of the other configurations (profiles/collect_matrix_stats.sh r04)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c15
t0=$(date +%s)
timeout -k 10 600 python bench.py --gpus 2 --share-device --dist-backend gloo --steps 20 --warmup 5 > gpurun_out/r04c15/n2.json 2> gpurun_out/r04c15/n2.err || { tail -20 gpurun_out/r04c15/n2.err; exit 1; }
echo "N = 2 rehearsal: $(( $(date +%s) - t0 )) s wall"
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r04c15/n2.json')); v=d['config']['variants']['configs3_rmat23']
print('n_gpus', d['n_gpus'], 'value %.3g' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'host', d['config']['host'], '| configs[3]: n_gpus', v['n_gpus'], 'value %.3g' % v['value'], 'ms/step %.4f' % v['ms_per_step'], 'parity', v['parity_max_err_vs_fp64_oracle'], v['timed_path_check']['bit_identical_to_oracle_gated_forward'])
PY
for i in $(seq 12); do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c15/a$i.json 2>/dev/null
  timeout -k 10 200 python bench.py --cpu-seconds 0 --no-variant --no-parity --scale-variant off > gpurun_out/r04c15/b$i.json 2>/dev/null
done
python3 - <<'PY'
import json, glob, statistics as st
for tag, name in (("a", "20 steps after 5"), ("b", "200 steps after 20")):
    v = [json.load(open(f))["ms_per_step"] * 1e3 for f in sorted(glob.glob(f"gpurun_out/r04c15/{tag}[0-9]*.json"))]
    print(f"{name}: n = {len(v)}, mean {st.mean(v):.2f} us, stdev {st.pstdev(v):.2f}, min {min(v):.2f}, max {max(v):.2f}")
PY
bash profiles/collect_matrix_stats.sh r04 2>&1 | tail -12
