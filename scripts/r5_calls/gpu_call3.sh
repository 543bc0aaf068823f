#!/usr/bin/env bash
# round 5, GPU call 3: the whole GPU suite on the final library + the stream-choice measurement
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== pytest -m gpu (whole suite)"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 | tee "$OUT/call3_pytest_gpu.txt"
echo "== stream choice"
timeout -k 10 600 python scripts/stream_choice.py 2>&1 | tee "$OUT/stream_choice.txt"
