#!/bin/bash
# r03 call 16: row ranges of the reproducible weight gradients (SAGE_BWD_DIRECT_BLOCKS) against the training step at config-3 size
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in 128 192 256 384 512; do echo "== SAGE_BWD_DIRECT_BLOCKS=$v"; SAGE_BWD_DIRECT_BLOCKS=$v timeout -k 10 300 python experiments/train_big.py 2>&1 | grep -E "ms per|captured"; done
