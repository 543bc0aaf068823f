#!/usr/bin/env bash
# round 5, GPU call 24: where a step of the resident dp_search goes (diagnostic library with s_memrealtime stamps)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
CPPFLOW_HIP_LIB=build_var/lib_dpstamp.so timeout -k 10 300 python scripts/dp_step_timeline.py 2>&1 | grep -v amdgpu.ids | tee "$OUT/dp_step_timeline.txt"
CPPFLOW_HIP_LIB=build_var/lib_dpstamp.so timeout -k 10 300 python scripts/dp_step_timeline.py 96 256 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/dp_step_timeline.txt"
CPPFLOW_HIP_LIB=build_var/lib_dpstamp.so timeout -k 10 300 python scripts/dp_step_timeline.py 256 256 2>&1 | grep -v amdgpu.ids | tee -a "$OUT/dp_step_timeline.txt"
