#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c30
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/experiments/pipe_sweep.py --steps 40 --warmup 10 --baseline 1 --bstreams 1 --order degree --configs > $O/t.log 2>&1
python3 $R/experiments/pipe_trace.py $O/t 2>&1 | grep -E "phase|^  +(So|Si|G|D|L2) n=" | cut -c1-100; rm -rf $O/t
cd $R
show() { python3 -c "
import json,sys;d=json.load(open('$1'));r=d['roofline'];print('$2',round(d['value']/1e6,2),'Memb/s',d['ms_per_step'],'ms fwd_frac',r['forward_frac'],'k_ms',r['kernel_ms'],'alone',r['kernel_ms_alone'], r['stage_ms_alone'])"; }
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/b200.json 2> $O/b200.err; show $O/b200.json "default 200/20:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/b20.json 2> $O/b20.err; show $O/b20.json "20/5:"
SAGE_SAMPLE_FUSED=0 timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/b200u.json 2> $O/b200u.err; show $O/b200u.json "two-launch sampler 200/20:"
