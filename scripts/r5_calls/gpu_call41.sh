#!/usr/bin/env bash
# round 5, GPU call 41: wall time of the driver's bench command on a fresh box (first import included)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
t0=$(date +%s.%N); python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_walltime.json" 2> "$OUT/bench_walltime.err"; echo "driver flags: wall $(python -c "import time; print(round(time.time() - $t0, 1))") s"
python -c "import json; d=json.load(open('$OUT/bench_walltime.json')); print('us/step %.2f value %.3e' % (d['ms_per_step']*1e3, d['value']))"
t0=$(date +%s.%N); python bench.py > "$OUT/bench_walltime2.json" 2> "$OUT/bench_walltime2.err"; echo "default flags: wall $(python -c "import time; print(round(time.time() - $t0, 1))") s"
