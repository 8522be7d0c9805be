#!/bin/bash
# r03 call 2: (a) capture patterns 8-12, (b) the new round-3 GPU test (role pipeline at config-3 size vs the oracle), (c) bench default
# 20/5 with the timed-path check, (d) TG / blocks of the concat contraction
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c2; mkdir -p $O
PATTERNS="8 9 10 11 12" bash experiments/r03/capture_repro.sh > $O/capture.log 2>&1; grep -E "pattern|exit|OK|WRONG" $O/capture.log
echo "=== tests"; timeout -k 10 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -s 2>&1 | tail -8
echo "=== bench 20/5"; timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_20_5.json 2> $O/bench_20_5.err || tail -5 $O/bench_20_5.err
python3 -c "
import json; d=json.load(open('$O/bench_20_5.json')); print(d['value'], d['ms_per_step'], d['timed_path_check'], d['roofline']['forward_frac'], d['roofline'].get('traffic_source'))"
