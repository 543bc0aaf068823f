#!/usr/bin/env bash
# round 5, GPU call 28: the per-waypoint exchange with all workgroups on one die, plain stores + sc1 loads (microbenchmark)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 300 build_var/dp_exchange 2>&1 | tee "$OUT/dp_exchange.txt"
