#!/usr/bin/env bash
# round 5, GPU call 4: the two-ranks-on-one-GPU product test, then the profiler passes of the record pass (counters, kernel trace, overlap)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_two_ranks_one_gpu.py tests/test_gpu_sharded_refiner.py -m gpu -q -x 2>&1 | tail -15 | tee "$OUT/call4_pytest.txt"
bash scripts/record_pass.sh pmc
