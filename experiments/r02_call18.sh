#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c18
mkdir -p $O
cd $R
show() { python3 -c "
import json,sys;d=json.load(open('$1'));r=d['roofline'];print('$2',round(d['value']/1e6,2),'Memb/s',d['ms_per_step'],'ms fwd_frac',r['forward_frac'],'k_ms',r['kernel_ms'],'alone',r['kernel_ms_alone'],'frac',r['frac'],r['frac_alone'], d['config']['preheat'][:12])"; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/b20.json 2> $O/b20.err; echo "rc=$?"; show $O/b20.json "20/5 default:"
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 > $O/b200.json 2> $O/b200.err; show $O/b200.json "200/20 default:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --depth 5 > $O/b20d5.json 2> $O/b20d5.err; show $O/b20d5.json "20/5 depth5:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --engine-layout input > $O/b20in.json 2> $O/b20in.err; show $O/b20in.json "20/5 layout=input:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --exec replay --streams 2 > $O/b20rep.json 2> $O/b20rep.err; show $O/b20rep.json "20/5 replay x2:"
timeout -k 10 300 python bench.py --steps 200 --warmup 20 --cpu-seconds 0 --exec replay --streams 2 > $O/b200rep.json 2> $O/b200rep.err; show $O/b200rep.json "200/20 replay x2:"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --preheat-seconds 0 > $O/b20nopre.json 2> $O/b20nopre.err; show $O/b20nopre.json "20/5 no preheat:"
timeout -k 10 300 python bench.py --gpus 2 --share-device --dist-backend gloo --steps 20 --warmup 5 --cpu-seconds 0 > $O/b_n2.json 2> $O/b_n2.err; echo "n2 rc=$?"; python3 -c "
import json;d=json.load(open('$O/b_n2.json'));print('N=2 self-launched (gloo, one shared GPU):',d['n_gpus'],d['value'],d['ms_per_step'])"
