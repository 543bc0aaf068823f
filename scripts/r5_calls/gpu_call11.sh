#!/usr/bin/env bash
# round 5, GPU call 11: the principal-range fast path of the lean residual's angle functions: whole suite, lean parity, A/B (lib_norpy = without)
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 1100 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee "$OUT/call11_pytest_gpu.txt"
echo "== A/B rpy fast path, default flags (2 000 steps)" | tee "$OUT/ab_lean_rpy_fast.txt"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_norpy.so 2>&1 | tee -a "$OUT/ab_lean_rpy_fast.txt"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_norpy.so; do
  echo -n "$lib  " | tee -a "$OUT/ab_lean_rpy_fast.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$OUT/ab_lean_rpy_fast.txt"
done; done
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_norpy.so; do for cfg in "--config C3" "--inputs random" "--config C5 --steps 200 --warmup 20"; do
  echo -n "$lib $cfg  " | tee -a "$OUT/ab_lean_rpy_fast.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us  conv %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['converged_frac_pos_err_lt_1e-4']))" | tee -a "$OUT/ab_lean_rpy_fast.txt"
done; done
