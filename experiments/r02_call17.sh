#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c17
mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --steps 200 --warmup 20 > $O/bench_default.json 2> $O/bench_default.err; echo "bench default rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_default.json'));r=d['roofline'];print(d['value'],d['ms_per_step'],'fwd_frac',r['forward_frac'],'k_ms',r['kernel_ms'],'alone',r['kernel_ms_alone'],'frac',r['frac'],r['frac_alone'],r['stage_ms_alone'],d['parity_max_err_vs_fp64_oracle'],d['cpu_baseline'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_20.json 2> $O/bench_20.err; echo "bench 20/5 rc=$?"; python3 -c "
import json;d=json.load(open('$O/bench_20.json'));print(d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
one() { tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 240 --warmup 24 --order degree --baseline 0 --tag $tag "$@" > $O/$tag.log 2>&1
  echo "$tag [G$SAGE_G_VARIANT G@$SAGE_G_PER_CU D@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS] $(grep 'us/forward' $O/$tag.log | sed -E 's/ +/ /g' | cut -c1-52 | tr '\n' ';')"
}
one base --configs 4:SGDL: 3:SGDL: 5:SGDL: 6:SGDL: 4:SGLD: 4:SGDD:
SAGE_G_PER_CU=5 one g5 --configs 4:SGDL: 5:SGDL:
SAGE_G_PER_CU=7 one g7 --configs 4:SGDL: 5:SGDL:
SAGE_DENSE_BLOCKS=192 one d192 --configs 4:SGDL:
SAGE_DENSE_BLOCKS=224 one d224 --configs 4:SGDL:
SAGE_T16_WAVES=8 one t8 --configs 4:SGDL:
SAGE_SO_THREADS=512 one so512 --configs 4:SGDL:
SAGE_G_VARIANT=1 SAGE_G_ROWS=2 SAGE_G_PER_CU=4 one r2g4 --configs 4:SGDL:
SAGE_G_VARIANT=1 SAGE_G_ROWS=2 SAGE_G_PER_CU=5 one r2g5 --configs 4:SGDL:
