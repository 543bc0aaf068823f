#!/usr/bin/env bash
# round 5, GPU call 2: the lean-iteration parity test with every bar evaluated (no -x)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q -s 2>&1 | grep -E "^E  |K=[235]: calm|passed|failed" | cut -c1-600 | tee "$OUT/call2_lean_pytest.txt" || true
