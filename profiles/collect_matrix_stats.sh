#!/bin/bash
# Per-kernel rocprofv3 summaries of the OTHER configurations, in the running pipeline and with one batch in flight (VERDICT r2 #8:
# every r0N_matrix_* line's dominant-kernel fraction must be recomputable):  gpurun -- 'bash profiles/collect_matrix_stats.sh r03'
# -> gpurun_out/<tag>_mstats/<name>[_alone]/..., reduced into profiles/<tag>_kernel_stats_<name>[_one_batch_in_flight].csv
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/${TAG}_mstats; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
run(){ name=$1; shift
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/$name" -- python3 "$R/bench.py" --steps 100 --cpu-seconds 0 --no-variant --scale-variant off "$@" \
      > "$O/$name.json" 2> "$O/$name.log"; echo "$name rc=$?"
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/${name}_alone" -- python3 "$R/bench.py" --steps 60 --warmup 10 --exec direct --streams 1 \
      --no-parity --cpu-seconds 0 --no-variant --scale-variant off "$@" > "$O/${name}_alone.json" 2> "$O/${name}_alone.log"; echo "${name}_alone rc=$?"
}
run c3_concat --config 3 --mode concat
run c4_rmat23 --config 4
run c5_gcn --config 5
run c5_concat --config 5 --mode concat
python3 "$R/profiles/reduce_matrix_stats.py" "$O" "$TAG"     # (run it again in the build container: only gpurun_out/ travels back)
find "$O" -name "*_kernel_trace.csv" -size +8M -delete
