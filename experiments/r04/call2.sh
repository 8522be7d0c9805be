#!/bin/bash
# role S serving two (four) batches per sampler launch; inner-hop grid capped (grid-stride over the worst-case list)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r04c2
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "pipe or sampl or frontier" > gpurun_out/r04c2/tests.log 2>&1 || { tail -30 gpurun_out/r04c2/tests.log; exit 1; }
tail -2 gpurun_out/r04c2/tests.log
SETTINGS=("SAGE_PIPE_PAIR=1 SAGE_SI_GRID=0" "SAGE_PIPE_PAIR=1 SAGE_SI_GRID=2048" "SAGE_PIPE_PAIR=2 SAGE_SI_GRID=2048" "SAGE_PIPE_PAIR=2 SAGE_SI_GRID=2048 SAGE_DEPTH=6" \
  "SAGE_PIPE_PAIR=2 SAGE_SI_GRID=2048 SAGE_DEPTH=8" "SAGE_PIPE_PAIR=4 SAGE_SI_GRID=2048 SAGE_DEPTH=8" "SAGE_PIPE_PAIR=1 SAGE_SI_GRID=1024" "SAGE_PIPE_PAIR=2 SAGE_SI_GRID=1024 SAGE_DEPTH=6")
echo "== 300 steps after 50"
STEPS=300 bash experiments/env_run.sh 2 "${SETTINGS[@]}" 2>&1 | cut -c1-200 | tee gpurun_out/r04c2/long.log
echo "== 20 steps after 5"
cat > /tmp/short_run.sh <<'EOS'
cd ${GRAFT_REPO_ROOT:-/root/repo}; O=gpurun_out/env20; mkdir -p $O
reps=$1; shift
for rep in $(seq $reps); do i=0; for e in "$@"; do i=$((i+1))
  env $e timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 --no-variant --no-parity > $O/e$i.$rep.json 2> $O/e$i.$rep.err || { echo "$e FAILED"; tail -3 $O/e$i.$rep.err; exit 1; }
  python3 -c "
import json; d=json.load(open('$O/e$i.$rep.json'))
print('rep $rep %5.1f us/fwd | %s' % (1e3*d['ms_per_step'], '$e'))"
done; done
EOS
bash /tmp/short_run.sh 3 "${SETTINGS[@]}" 2>&1 | tee gpurun_out/r04c2/short.log
