#!/usr/bin/env bash
# round 5, GPU call 6: launches in flight at FULL size (the headline workload): 2 / 3 / 4 streams, driver flags and 2 000 steps
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
rm -f "$OUT/fullsize_streams.txt"
for rep in 1 2; do for st in 2 3 4; do for steps in 20 2000; do
  w=5; [ $steps = 2000 ] && w=100
  timeout -k 10 200 python bench.py --gpus 1 --steps $steps --warmup $w --streams $st --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('streams $st steps $steps: us/step %.2f  (min %.2f max %.2f)' % (d['ms_per_step']*1e3, d['config']['timed_region']['ms_per_step_min']*1e3, d['config']['timed_region']['ms_per_step_max']*1e3))" | tee -a "$OUT/fullsize_streams.txt"
done; done; done
