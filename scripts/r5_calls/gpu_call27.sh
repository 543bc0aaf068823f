#!/usr/bin/env bash
# round 5, GPU call 27: step timeline of the table + resident hand-off kernel
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
CPPFLOW_HIP_LIB=build_var/lib_dpstamp_t.so timeout -k 10 300 python scripts/dp_step_timeline_tabled.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dp_step_timeline_tabled.txt"
