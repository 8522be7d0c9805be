#!/bin/bash
# the RCCL calls of bench.py with one rank; the two-rank line test; N = 4 rehearsal on loopback-pinned gloo (progress marks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
set -o pipefail
timeout -k 10 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -x -q -m gpu -k "rccl or two_ranks" 2>&1 | tail -6 &&
bash experiments/r04/call34.sh 4
