#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02c4
mkdir -p $O
cd $R
export SAGE_G_VARIANT=1 SAGE_G_PER_CU=4 SAGE_DENSE_VARIANT=1 SAGE_DENSE_BLOCKS=256 SAGE_T16_WAVES=8 SAGE_SO_THREADS=256
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_ops.py -x -q > $O/pytest_newkernels.log 2>&1; echo "pytest(new kernels) rc=$?"; tail -3 $O/pytest_newkernels.log
cd /tmp && export TMPDIR=/tmp
run() {  # tag, configs...
  tag=$1; shift
  timeout -k 10 300 python3 $R/experiments/pipe_sweep.py --steps 200 --warmup 20 --baseline 0 --tag $tag --configs "$@" > $O/$tag.log 2>&1
  echo "== $tag rc=$?  [G$SAGE_G_VARIANT@$SAGE_G_PER_CU D$SAGE_DENSE_VARIANT@$SAGE_DENSE_BLOCKS T16w$SAGE_T16_WAVES So$SAGE_SO_THREADS]"; grep "us/forward" $O/$tag.log | cut -c1-75
}
run B 4:SGDL: 3:SGDD: 4:SGGL: 6:SGDL: 1:SSSS:
SAGE_SO_THREADS=512 run C 4:SGDL: 3:SGDD:
SAGE_G_PER_CU=5 run E5 4:SGDL:
SAGE_G_PER_CU=6 run E6 4:SGDL:
SAGE_DENSE_BLOCKS=128 run F128 4:SGDL:
SAGE_DENSE_VARIANT=0 SAGE_DENSE_BLOCKS=128 run Dold128 4:SGDL: 4:SGGL:
SAGE_T16_WAVES=16 run L16 4:SGDL:
# timeline of the best-guess config + alone durations
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace_B -- python3 $R/experiments/pipe_sweep.py --steps 100 --warmup 20 --baseline 1 --bstreams 1 --configs 4:SGDL: > $O/trace_B.log 2>&1
python3 $R/experiments/pipe_trace.py $O/trace_B > $O/trace_B.txt 2>&1
grep -E "phase|^  +(So|Si|G|D|L2) n=|gap|running" $O/trace_B.txt | cut -c1-150
rm -rf $O/trace_B
