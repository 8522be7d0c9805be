#!/bin/bash
# The OPTIONAL producer / consumer contraction (SAGE_DENSE_PC=1, csrc/sage_dense.hip: dense_pc_kernel) under the profiler, one batch in
# flight: kernel stats and the matrix-pipe counters, for comparison with the default kernel's (profiles/mfma.json).
#   gpurun -- 'bash profiles/collect_dense_pc.sh'   then   python profiles/reduce.py gpurun_out/r03pc r03pc
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03pc; mkdir -p "$O"
export SAGE_DENSE_PC=1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_alone" -- python3 "$R/bench.py" --exec direct --streams 1 \
    --steps 100 --warmup 10 --no-parity --cpu-seconds 0 --no-variant > "$O/bench_alone_under_rocprof.json" 2> "$O/stats_alone.log"; echo "stats_alone rc=$?"
PMC_CMD="python3 $R/bench.py --steps 12 --warmup 3 --exec direct --streams 1 --no-parity --cpu-seconds 0 --preheat-seconds 0"
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
             "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    name=pmc_$(echo $group | tr ' ' '_' | cut -c1-48)
    timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$O/$name" -- $PMC_CMD > "$O/$name.log" 2>&1; echo "$name rc=$?"
done
python3 "$R/profiles/reduce.py" "$O" r03pc
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
find "$O" -name "*_counter_collection.csv" -size +8M -delete
