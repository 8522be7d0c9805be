#!/bin/bash
# r03 call 6: the role pipeline captured as one hipGraph: test, then bench pipe vs pipegraph in the driver's form (20 / 5) and at 200 / 20
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/r03c6; mkdir -p $O
timeout -k 10 600 python -X faulthandler -m pytest tests/test_gpu_round3.py -x -q -m gpu -s -k "captured_as_one_graph" > $O/tests.log 2>&1; tail -15 $O/tests.log
grep -q "passed" $O/tests.log || exit 1
for rep in 1 2 3; do for mode in pipe pipegraph; do for sw in "20 5" "200 20"; do set -- $sw
  timeout -k 10 300 python bench.py --exec $mode --steps $1 --warmup $2 --cpu-seconds 0 --no-variant > $O/$mode.$1.$rep.json 2> $O/$mode.$1.$rep.err || { echo "$mode $1 FAILED"; tail -5 $O/$mode.$1.$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/$mode.$1.$rep.json')); print('%-9s steps %3d rep $rep: %.2f us/fwd  host %.1f us  check %s' % ('$mode', $1, 1e3*d['ms_per_step'], 1e3*d['config']['host_enqueue_ms_per_step'], d['timed_path_check']['bit_identical_to_oracle_gated_forward']))"
done; done; done
