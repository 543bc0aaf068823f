#!/usr/bin/env bash
# round 5, GPU call 7: the lean angle functions (one shared reciprocal, shorter polynomials) + the one-instruction clamp: whole suite, then
# the A/B against the old forms on this one box (in-tree = both on; lib_old = both off; lib_trig / lib_med3 = one each)
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== pytest -m gpu (whole suite)"
timeout -k 10 1100 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee "$OUT/call7_pytest_gpu.txt"
echo "== A/B, default flags (2 000 steps)"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_old.so build_var/lib_trig.so build_var/lib_med3.so 2>&1 | tee "$OUT/ab_lean_trig.txt"
echo "== A/B, driver flags"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_old.so; do
  echo -n "$lib  " | tee -a "$OUT/ab_lean_trig.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$OUT/ab_lean_trig.txt"
done; done
