#!/bin/bash
# round 2, GPU call 1: GPU tests, role-pipeline sweep, MFMA counters of the existing kernels
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c1
mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout -k 10 400 python experiments/pipe_sweep.py --steps 200 --warmup 20 > $O/sweep_default.log 2>&1; echo "sweep rc=$?" | tee -a $O/summary.txt
grep -E "us/forward" $O/sweep_default.log | tee -a $O/summary.txt
SAGE_G_PER_CU=6 SAGE_SO_THREADS=256 timeout -k 10 300 python experiments/pipe_sweep.py --steps 200 --warmup 20 --baseline 0 --tag g6_so256 > $O/sweep_g6.log 2>&1
grep -E "us/forward" $O/sweep_g6.log | tee -a $O/summary.txt
SAGE_G_PER_CU=4 SAGE_SO_THREADS=256 timeout -k 10 300 python experiments/pipe_sweep.py --steps 200 --warmup 20 --baseline 0 --tag g4_so256 > $O/sweep_g4.log 2>&1
grep -E "us/forward" $O/sweep_g4.log | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
PMC_CMD="python3 $R/bench.py --steps 10 --warmup 2 --streams 1 --no-graph --no-parity --cpu-seconds 0 --node-order original"
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC"; do
    name=pmc_$(echo $group | tr ' ' '_' | cut -c1-60)
    timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d $O/$name -- $PMC_CMD > $O/$name.log 2>&1
    echo "$name rc=$?" | tee -a $O/summary.txt
done
echo done | tee -a $O/summary.txt
