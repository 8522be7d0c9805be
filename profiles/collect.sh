#!/bin/bash
# Regenerates the round's evidence on an MI355X box:  gpurun -- 'bash profiles/collect.sh r02'
# Writes gpurun_out/<tag>/...; profiles/reduce.py turns that into the files committed under profiles/.
# PMC passes are separate runs with no trace domains (gpurun refuses --pmc combined with sys/hip/hsa tracing); counters are
# collected with the forward host-enqueued on ONE stream (rocprofv3 serialises dispatches under --pmc anyway).
set -o pipefail
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
timeout -k 10 400 python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > "$O/bench_20_5.json" 2> "$O/bench_20_5.err"
timeout -k 10 300 python bench.py --cpu-seconds 0 --scale-variant off --engine-layout input > "$O/bench_layout_input.json" 2> "$O/bench_layout_input.err"
timeout -k 10 300 python bench.py --cpu-seconds 0 --scale-variant off --exec replay --streams 2 > "$O/bench_replay2.json" 2> "$O/bench_replay2.err"
cd /tmp && export TMPDIR=/tmp
# (--no-variant: the variants' pipelines -- caller's node order, pre-transformed table -- launch the same kernels on other data and
#  would be averaged into the per-kernel summary)
rm -rf "$O/stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --cpu-seconds 0 --no-variant --scale-variant off \
    > "$O/bench_under_rocprof.json" 2> "$O/stats.log"; echo "stats rc=$?"
# the same kernels with ONE batch in flight (host-enqueued on one stream): what each costs alone
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats_alone" -- python3 "$R/bench.py" --exec direct --streams 1 \
    --steps 100 --warmup 10 --no-parity --cpu-seconds 0 --no-variant --scale-variant off > "$O/bench_alone_under_rocprof.json" 2> "$O/stats_alone.log"; echo "stats_alone rc=$?"
PMC_CMD="python3 $R/bench.py --steps 12 --warmup 3 --exec direct --streams 1 --no-parity --cpu-seconds 0 --preheat-seconds 0 --scale-variant off"
for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES" \
             "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"; do
    name=pmc_$(echo $group | tr ' ' '_' | cut -c1-48)
    timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$O/$name" -- $PMC_CMD > "$O/$name.log" 2>&1; echo "$name rc=$?"
done
# the contraction in isolation at BASELINE configs[4] (D0 = 100, H = 128: the MFMA-path configuration): matrix-pipe counters
PMC5="python3 $R/bench.py --config 5 --steps 12 --warmup 3 --exec direct --streams 1 --no-parity --cpu-seconds 0 --preheat-seconds 0"
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES"; do
    name=c5_pmc_$(echo $group | tr ' ' '_' | cut -c1-48)
    timeout -k 10 400 rocprofv3 --pmc $group --output-format csv -d "$O/$name" -- $PMC5 > "$O/$name.log" 2>&1; echo "$name rc=$?"
done
python3 "$R/profiles/reduce.py" "$O" "$TAG"
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
find "$O" -name "*_counter_collection.csv" -size +8M -delete
