#!/usr/bin/env bash
# round 5, GPU call 22: 150 s soak of the final library (fused launches on four streams with resident dp_search runs in between, every result compared)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 400 python scripts/soak.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/soak.txt"
