#!/bin/bash
# bench lines of the other configurations -> gpurun_out/matrix/*.json   (experiments/matrix.sh [steps] [tag])
S=${1:-100}
T=${2:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/matrix; mkdir -p $O
cd $R
run(){ name=$1; shift; timeout -k 10 500 python bench.py --steps $S --cpu-seconds 0 --no-variant --scale-variant off "$@" > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; return 0; }
  python3 -c "
import json; d=json.load(open('$O/$name.json')); r=d['roofline']
print('$name', 'us/fwd %.1f' % (1e3*d['ms_per_step']), 'emb/s %.3g' % d['value'], 'parity %.1e' % d['parity_max_err_vs_fp64_oracle'], 'fwd_frac', r['forward_frac'], 'frac', r['frac'], 'alone', r['frac_alone'], {k: round(v*1e3,1) for k,v in r['stage_ms_alone'].items()})"; }
run ${T}_matrix_c2_pubmed --config 2
run ${T}_matrix_c2_pubmed_concat --config 2 --mode concat
run ${T}_matrix_c3_concat --config 3 --mode concat
run ${T}_matrix_c3_selfloop --config 3 --self-loop
run ${T}_matrix_c4_rmat23 --config 4
run ${T}_matrix_c5_gcn --config 5
run ${T}_matrix_c5_concat --config 5 --mode concat
run ${T}_matrix_c5_selfloop --config 5 --self-loop
