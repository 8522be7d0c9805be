#!/bin/bash
# EXPERIMENT: inner hop writes every node's sampled ids hot-first (id < T in the degree order); the gather requests list positions >= U with streaming (nt) loads
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c16
STEPS=300 bash experiments/ab_run.sh 2 base hf16 hf16_nt6 hf16_nt9 hf16_nt12 hf64_nt6 hf64_nt9 hf64_nt12 hf4_nt4 2>&1 | cut -c1-150 | tee gpurun_out/r04c16/ab.log
