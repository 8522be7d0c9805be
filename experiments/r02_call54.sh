#!/bin/bash
# sweep gather with the block's waves in lock step: bytes past L2 and time
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/sweep; mkdir -p $O
export TMPDIR=/tmp
for lib in mb_sweep.so mb_sweep_ls.so; do for mode in 0,256 1,512; do
  export MB_LIB=$lib MB_ONLY=$mode MB_BOUNDS=mass
  echo "== $lib mode $mode"; timeout -k 10 300 python experiments/mb_sweep.py 2>&1 | grep -E "sweep mode|product"
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/pmc54 -o x --output-format csv -- python3 experiments/mb_sweep.py > $O/pmc54.log 2>&1 || { echo "pmc failed"; tail -5 $O/pmc54.log; exit 1; }
  python3 experiments/pmc_gather.py $O/pmc54 gm_sweep; rm -rf $O/pmc54
done; done
