#!/bin/bash
# bench lines of the other configurations -> gpurun_out/matrix/*.json   (experiments/matrix.sh [steps])
S=${1:-100}
O=$GRAFT_REPO_ROOT/gpurun_out/matrix; mkdir -p $O
run(){ name=$1; shift; timeout -k 10 400 python bench.py --steps $S --cpu-seconds 0 "$@" > $O/$name.json 2> $O/$name.err || { tail -3 $O/$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('$O/$name.json')); r=d['roofline']
print('$name', 'us/fwd %.1f' % (1e3*d['ms_per_step']), 'emb/s %.3g' % d['value'], 'parity %.1e' % d['parity_max_err_vs_fp64_oracle'], 'fwd_frac', r['forward_frac'], r['kernel'][:24], 'frac', r['frac'], {k: round(v*1e3,1) for k,v in r['stage_ms'].items()})"; }
run r01_matrix_c2_pubmed --config 2 && run r01_matrix_c3_gcn --config 3 && run r01_matrix_c3_concat --config 3 --mode concat && run r01_matrix_c5_gcn --config 5 && run r01_matrix_c5_concat --config 5 --mode concat && run r01_matrix_c5_selfloop --config 5 --self-loop
