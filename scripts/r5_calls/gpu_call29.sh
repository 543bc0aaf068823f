#!/usr/bin/env bash
# round 5, GPU call 29: final confirmation on the committed tree: the GPU suite, the driver's smoke() hook, the driver's bench command
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pytest_gpu_final.txt"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tee "$OUT/bench_final_driverflags.json" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags: us/step %.2f value %.3e frac %.3f one_stream %.2f random %.2f cpu %s' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac'], d['one_stream']['ms_per_step']*1e3, d['random_inputs']['ms_per_step']*1e3, d['cpu_baseline']['value']))"
