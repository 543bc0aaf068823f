#!/usr/bin/env bash
# round 5, GPU call 8: final library (lean angle functions in): whole suite, lean-parity statistics, profiler passes of the record pass
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee "$OUT/call8_pytest_gpu.txt"
timeout -k 10 600 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q -s 2>&1 | grep -E "^E  |K=[235]: calm|passed|failed" | cut -c1-400 | tee "$OUT/call8_lean_pytest.txt"
timeout -k 10 600 python scripts/lean_parity_stats.py > "$OUT/lean_parity.txt" 2> "$OUT/lean_parity.err"
bash scripts/record_pass.sh pmc
