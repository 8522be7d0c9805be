#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c19
mkdir -p $O
cd $R
show() { python3 -c "
import json,sys;d=json.load(open('$1'));r=d['roofline'];print('$2',round(d['value']/1e6,2),'Memb/s',d['ms_per_step'],'ms fwd_frac',r['forward_frac'],'k_ms',r['kernel_ms'],'alone',r['kernel_ms_alone'])"; }
for q in 2 4 6 8 12; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-parity > $O/q$q.json 2> $O/q$q.err; show $O/q$q.json "pipe SGDL d4, GPU_MAX_HW_QUEUES=$q:"
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-parity --depth 5 > $O/q8d5.json 2> $O/q8d5.err; show $O/q8d5.json "pipe d5 q8:"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-parity --roles SGDD > $O/q8sgdd.json 2> $O/q8sgdd.err; show $O/q8sgdd.json "pipe SGDD q8:"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-parity > $O/q8_20.json 2> $O/q8_20.err; show $O/q8_20.json "pipe 20/5 q8:"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-parity --exec replay --streams 2 > $O/q8rep.json 2> $O/q8rep.err; show $O/q8rep.json "replay x2 q8:"
GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --no-parity --exec replay --streams 3 > $O/q8rep3.json 2> $O/q8rep3.err; show $O/q8rep3.json "replay x3 q8:"
