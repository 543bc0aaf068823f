#!/usr/bin/env bash
# round 5, GPU call 30: dp_search resident kernels publishing a workgroup's four results with ONE store instruction: parity, fuzz, A/B
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_dp_publish1.txt"; : > "$F"
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "dp or search or plan" 2>&1 | tail -4 | tee "$OUT/call30_pytest.txt"
timeout -k 10 300 python scripts/fuzz_dp.py 2>&1 | tail -2 | tee -a "$F"
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_pub4.so cppflow_amd/csrc/libcppflow_hip.so build_var/lib_pub4.so; do
  echo "== $lib" | tee -a "$F"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 300 python scripts/dp_bench.py 2>&1 | grep -E "k=  175 T= 256|k=  300|k=  512|k= 1024|k=  256|k=  128|k=  175 T=  64" | cut -c1-40,95-250 | tee -a "$F"
done
