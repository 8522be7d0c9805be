#!/bin/bash
# L2 hit / beyond-L2 reads of the sweep gather with equal-mass buckets vs uniform id ranges vs the product gather
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/sweep; mkdir -p $O
export TMPDIR=/tmp MB_ONLY=0,256
for b in mass uni; do
  export MB_BOUNDS=$b
  timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $O/pmc_$b -o x --output-format csv -- python3 experiments/mb_sweep.py > $O/pmc_$b.log 2>&1 || { echo "pmc $b failed"; tail -5 $O/pmc_$b.log; exit 1; }
  echo "== $b: sweep"; python3 experiments/pmc_gather.py $O/pmc_$b gm_sweep
  echo "== $b: product"; python3 experiments/pmc_gather.py $O/pmc_$b gather_mean
done
