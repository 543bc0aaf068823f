#!/usr/bin/env bash
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k pacing 2>&1 | tail -40 | cut -c1-220 | tee "$OUT/call32_pytest.txt"
