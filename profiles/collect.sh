#!/bin/bash
# Regenerates the round's evidence on an MI355X box:  gpurun -- 'bash profiles/collect.sh r01'
# Writes gpurun_out/<tag>/...; profiles/reduce.py turns that into the files committed under profiles/.
# PMC passes are separate runs with no trace domains (gpurun refuses --pmc combined with sys/hip/hsa tracing).
set -eo pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R"
timeout -k 10 300 python bench.py > "$O/bench.json" 2> "$O/bench.err"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/stats" -- python3 "$R/bench.py" --cpu-seconds 0 \
    > "$O/bench_under_rocprof.json" 2> "$O/stats.log"
PMC_CMD="python3 $R/bench.py --steps 10 --warmup 2 --streams 1 --no-graph --no-parity --cpu-seconds 0"
for group in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    name=pmc_$(echo $group | tr ' ' '_')
    timeout -k 10 300 rocprofv3 --pmc $group --output-format csv -d "$O/$name" -- $PMC_CMD > "$O/$name.log" 2>&1
done
python3 "$R/profiles/reduce.py" "$O" "$TAG"
