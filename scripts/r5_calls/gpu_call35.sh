#!/usr/bin/env bash
# round 5, GPU call 35: CPPF_TUNE_LM_PACE sweep on the product library
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 400 python scripts/pace_sweep.py panda 2>&1 | grep -v amdgpu.ids | tee "$OUT/pace_sweep.txt"
timeout -k 10 400 python scripts/pace_sweep.py fetch 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/pace_sweep.txt"
