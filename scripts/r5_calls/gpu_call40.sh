#!/usr/bin/env bash
# round 5, GPU call 40: experiment: the last iteration of a plain K >= 2 launch lean too (lib_lastlean.so): which parity tests notice, and what it buys
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_last_lean.txt"; : > "$F"
CPPFLOW_HIP_LIB=build_var/lib_lastlean.so timeout -k 10 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "^FAILED|passed|failed" | cut -c1-200 | tee -a "$F"
echo "== A/B" | tee -a "$F"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_lastlean.so 2>&1 | tee -a "$F"
