#!/bin/bash
# S and L (and S alone) on reserved CUs
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c1
timeout -k 10 900 python experiments/r04/cu_reserve.py base "S=0,L=0,G=1-7,D=1-7" "S=0+1,L=0+1,G=2-7,D=2-7" "S=0,L=1,G=2-7,D=2-7" "S=0,L=0" "S=0+1,L=0+1" \
   "S=0" "S=0+1" "S=0,G=1-7,D=1-7,L=1-7" "S=0+1,G=2-7,D=2-7,L=2-7" "S=0-3,L=0-3,G=4-7,D=4-7" base2 2>&1 | tee gpurun_out/r04c1/cu.log
