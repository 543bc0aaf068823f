#!/usr/bin/env bash
# round 5, GPU call 15: fair-share pacing experiment (diagnostic library build_var/lib_pace.so)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
CPPFLOW_HIP_LIB=build_var/lib_pace.so timeout -k 10 400 python scripts/pace_probe.py ${1:-} 2>&1 | grep -v amdgpu.ids | tee "$OUT/pace_probe.txt"
