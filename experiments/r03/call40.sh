#!/bin/bash
# launch tunables again, now that the host is off the critical path (one enqueue thread per role stream): which of round 2's / 3's
# "no effect" results were masked by the 50-us host loop?
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r03c40
STEPS=300 bash experiments/env_run.sh 1 "SAGE_X=0" "SAGE_SAMPLE_FUSED=1" "SAGE_G_PER_CU=4" "SAGE_G_PER_CU=5" "SAGE_G_PER_CU=8" "SAGE_T16_WAVES=16" "SAGE_T16_GRID=256" "SAGE_T16_GRID=768" \
  "SAGE_SO_THREADS=256" "SAGE_SO_THREADS=1024" "SAGE_ROLES=SGDD" "SAGE_ROLES=SSDL" "SAGE_ROLES=SGGL" "SAGE_TABLE_SLICE_FLOATS=64" "SAGE_G_VARIANT_SM=1" "SAGE_G_VARIANT_SM=1 SAGE_TABLE_SLICE_FLOATS=64" \
  "SAGE_TABLE_SLICED=0" "SAGE_X=0" 2>&1 | cut -c1-130 | tee gpurun_out/r03c40/knobs.log
