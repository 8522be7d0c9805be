#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c27
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
show() { python3 -c "
import json,sys;d=json.load(open('$1'));r=d['roofline'];print('$2',round(d['value']/1e6,2),'Memb/s',d['ms_per_step'],'ms fwd_frac',r['forward_frac'],'k_ms',r['kernel_ms'],'alone',r['kernel_ms_alone'],'frac',r['frac'],r['frac_alone'],'traffic',r['traffic'], 'mfma' in r)"; }
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/b200.json 2> $O/b200.err; show $O/b200.json "default 200/20:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/b20.json 2> $O/b20.err; show $O/b20.json "20/5:"
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/b200b.json 2> $O/b200b.err; show $O/b200b.json "default 200/20 again:"
