#!/usr/bin/env bash
# round 5, GPU call 12: final library (lean angle functions + principal-range fast path + block solve): lean-parity test and statistics, then
# the profiler passes of the record pass (counters, kernel trace, overlap trace)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q -s 2>&1 | grep -E "^E  |K=[235]: calm|passed|failed" | cut -c1-400 | tee "$OUT/call12_lean_pytest.txt"
timeout -k 10 600 python scripts/lean_parity_stats.py > "$OUT/lean_parity.txt" 2> "$OUT/lean_parity.err"
bash scripts/record_pass.sh pmc
