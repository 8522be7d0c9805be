#!/bin/bash
# rehearsal of the N = 4 launch (four ranks share this box's one GPU, gloo): host plan falls back to "submitting thread only" (16 cores / 4 ranks < 5),
# configs[3] rides along, closing_fence is reported
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c33
s=$(date +%s)
timeout -k 10 900 python bench.py --gpus 4 --share-device --dist-backend gloo --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r04c33/n4.json 2> gpurun_out/r04c33/n4.err || { tail -20 gpurun_out/r04c33/n4.err; exit 1; }
echo "wall $(( $(date +%s) - s )) s"
python3 -c "
import json; l=json.load(open('gpurun_out/r04c33/n4.json')); v=l['config']['variants']['configs3_rmat23']
print('N=4 on one GPU: value %.3e, %.1f us/step, host %s' % (l['value'], 1e3*l['ms_per_step'], l['config']['host']))
print('closing_fence', l['config']['closing_fence'])
print('checks', l['timed_path_check']['bit_identical_to_oracle_gated_forward'], v['timed_path_check']['bit_identical_to_oracle_gated_forward'], 'configs3 %.1f us/step' % (1e3*v['ms_per_step']), 'parity', l['parity_max_err_vs_fp64_oracle'], v['parity_max_err_vs_fp64_oracle'])"
