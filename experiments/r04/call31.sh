#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for t in 1 0; do timeout -k 10 300 python experiments/r04/express_debug.py 20 16000000 $t 2>&1 | grep -v amdgpu.ids || exit 1; done
SAGE_PIPE_EXPRESS=0 timeout -k 10 300 python experiments/r04/express_debug.py 20 16000000 1 2>&1 | grep -v amdgpu.ids
