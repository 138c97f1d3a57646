#!/bin/bash
# same-box A/B of the training step under two environments: tools/ab_env.sh "VAR=a" "VAR=b" [reps]
A=$1; B=$2; R=${3:-2}
for r in $(seq $R); do
  for e in "$A" "$B"; do
    out=$(env $e python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-torch-adam --no-probe $BENCH_ARGS 2>/dev/null | tail -1)
    echo "$e  $(echo "$out" | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.3f host %.3f eager %s" % (d["ms_per_step"], d["host_enqueue_ms_per_step"], d.get("ms_per_step_eager")))')"
  done
done
