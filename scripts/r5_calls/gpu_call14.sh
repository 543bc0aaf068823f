#!/usr/bin/env bash
# round 5, GPU call 14: the general lean angle path / wave-independence test, the driver's smoke() hook, one default bench line
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q 2>&1 | tail -12 | tee "$OUT/call14_pytest.txt"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags: us/step %.2f value %.3e frac %s' % (d['ms_per_step']*1e3, d['value'], d['roofline']['frac']))"
