#!/bin/bash
# role streams on CU subsets
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/cumask; mkdir -p $O
run(){ echo "== $1"; env $1 timeout -k 10 300 python experiments/cu_mask.py "${@:2}" 2>&1 | grep -E "us/forward|Error|error|rc=" ; }
run "X=0" "S=4,G=4,D=4,L=4" "G=3,D=-1" || exit 1
run "SAGE_DENSE_BLOCKS=64" "G=3,D=-1" "G=3,D=-1,L=-1" "G=3,D=-1,S=3"
run "SAGE_DENSE_BLOCKS=128" "G=2,D=-2" "G=3,D=-2"
run "SAGE_DENSE_BLOCKS=64 SAGE_G_PER_CU=4" "G=3,D=-1"
run "SAGE_DENSE_BLOCKS=64 SAGE_G_PER_CU=8" "G=3,D=-1"
