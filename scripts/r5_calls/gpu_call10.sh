#!/usr/bin/env bash
# round 5, GPU call 10: whole suite on the library with the lean block solve in; A/B of a wave-uniform skip of asin's half-angle form
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/call10_pytest_gpu.txt"
echo "== A/B asin branch, default flags (2 000 steps)" | tee "$OUT/ab_lean_asin_branch.txt"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_asinbr.so 2>&1 | tee -a "$OUT/ab_lean_asin_branch.txt"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_asinbr.so; do
  echo -n "$lib  " | tee -a "$OUT/ab_lean_asin_branch.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$OUT/ab_lean_asin_branch.txt"
done; done
