#!/bin/bash
# experiments/ab_build.sh <name> [-DFLAG ...]: a variant build of the product library -> experiments/ab/<name>.so (select with SAGE355_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd); name=$1; shift
mkdir -p $R/experiments/ab
cd $R/graphsage-simple_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-function "$@" -shared \
  sage_api.hip sage_sample.hip sage_gather.hip sage_linear.hip sage_fused.hip sage_dense.hip sage_forward.hip sage_backward.hip sage_backward_det.hip sage_pipe.hip \
  -o $R/experiments/ab/$name.so
echo built experiments/ab/$name.so
