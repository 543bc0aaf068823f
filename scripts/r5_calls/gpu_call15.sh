#!/usr/bin/env bash
# round 5, GPU call 15: fair-share pacing experiment (diagnostic libraries build_var/lib_pace*.so)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
SPEC="${1:-0,-300/4,-270/4,0,-300,0}"
CPPFLOW_HIP_LIB=build_var/lib_pace.so timeout -k 10 300 python scripts/pace_probe.py "$SPEC" 2>&1 | grep -v amdgpu.ids | tee "$OUT/pace_probe.txt"
echo "== with the finish-stage priority raise (lib_pace_h.so)" | tee -a "$OUT/pace_probe.txt"
CPPFLOW_HIP_LIB=build_var/lib_pace_h.so timeout -k 10 300 python scripts/pace_probe.py "$SPEC" 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/pace_probe.txt"
