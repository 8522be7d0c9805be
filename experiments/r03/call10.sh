#!/bin/bash
# r03 call 10: the whole GPU suite + smoke
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c10; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; tail -15 $O/tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
