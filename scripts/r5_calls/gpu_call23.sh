#!/usr/bin/env bash
# round 5, GPU call 23: dp_search with two polling re-reads in flight (A/B) + exactness fuzz on the variant
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_dp_poll2.txt"; : > "$F"
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_poll2_1.so build_var/lib_poll2_4.so cppflow_amd/csrc/libcppflow_hip.so build_var/lib_poll2_1.so; do
  echo "== $lib" | tee -a "$F"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 300 python scripts/dp_bench.py 2>&1 | grep -E "^panda.*(k=  175 T= 256|k=  300|k= 1024 T= 256|k=   64)" | cut -c1-200 | tee -a "$F"
done
echo "== exactness: scripts/fuzz_dp.py on lib_poll2_1.so" | tee -a "$F"
CPPFLOW_HIP_LIB=build_var/lib_poll2_1.so timeout -k 10 300 python scripts/fuzz_dp.py 2>&1 | tail -3 | tee -a "$F"
