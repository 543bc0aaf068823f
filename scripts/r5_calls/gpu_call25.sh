#!/usr/bin/env bash
# round 5, GPU call 25: dp_search off the transition table with the resident hand-off recurrence: parity, then the bench
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_dp_table_resident.py -m gpu -x -q 2>&1 | tail -12 | tee "$OUT/call25_pytest.txt"
timeout -k 10 600 python scripts/dp_bench.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dp_bench.txt"
