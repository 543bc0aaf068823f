#!/usr/bin/env bash
# round 5, GPU call 39: the developer fuzzers once more on the final library
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
: > "$OUT/fuzz_final.txt"
for f in fuzz_lm fuzz_masks fuzz_dp fuzz_coupled; do
  echo "== scripts/$f.py" | tee -a "$OUT/fuzz_final.txt"
  timeout -k 10 280 python scripts/$f.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-220 | tee -a "$OUT/fuzz_final.txt"
done
