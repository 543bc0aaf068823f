#!/usr/bin/env bash
# round 5, GPU call 20: the whole GPU suite on the final library (with the incoherent-rows mask tests)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest_gpu_final.txt"
