#!/bin/bash
# r03 call 14: the N > 1 form with the configs[3] variant, rehearsed with two ranks sharing this box's GPU (gloo): does it work, how long does it take?
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c14; mkdir -p $O
SECONDS=0; timeout -k 10 1000 python bench.py --gpus 2 --share-device --dist-backend gloo --steps 20 --warmup 5 --cpu-seconds 0 > $O/n2.json 2> $O/n2.err; echo "rc=$?"
echo "elapsed ${SECONDS}s"; tail -3 $O/n2.err
python3 -c "
import json; d=json.load(open('$O/n2.json')); print(d['n_gpus'], d['value'], d['ms_per_step'], d['timed_path_check'], json.dumps(d['config']['variants'])[:600])"
