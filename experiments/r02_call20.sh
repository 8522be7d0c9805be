#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c20
mkdir -p $O
cd $R
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log
