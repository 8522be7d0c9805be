#!/bin/bash
# bench.py's closing_fence fields: the 2-rank shared-GPU test + the driver's N = 1 form
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c27
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "two_ranks" 2>&1 | tail -5 &&
timeout -k 10 300 python bench.py --gpus 2 --share-device --dist-backend gloo --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --scale-variant off > gpurun_out/r04c27/n2.json 2> gpurun_out/r04c27/n2.err &&
python3 -c "
import json; l=json.load(open('gpurun_out/r04c27/n2.json')); print('N=2 shared GPU (gloo):', l['ms_per_step'], l['config']['closing_fence'], l['config']['host'])" &&
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04c27/n1.json 2> gpurun_out/r04c27/n1.err &&
python3 -c "
import json; l=json.load(open('gpurun_out/r04c27/n1.json')); print('N=1:', l['ms_per_step'], l['config']['closing_fence'], l['config']['variants']['configs3_rmat23']['ms_per_step'])"
