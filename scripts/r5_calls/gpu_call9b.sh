#!/usr/bin/env bash
# round 5, GPU call 9b: the block solve's lean-parity violations in full, and its A/B
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q -s 2>&1 | grep -E "^E  |K=[235]: calm|passed|failed" | cut -c1-700 | tee "$OUT/call9_lean_pytest.txt"
echo "== A/B, default flags (2 000 steps)" | tee "$OUT/ab_lean_block_solve.txt"
timeout -k 10 600 bash scripts/lib_ab.sh build_var/lib_noblk.so 2>&1 | tee -a "$OUT/ab_lean_block_solve.txt"
echo "== A/B, driver flags" | tee -a "$OUT/ab_lean_block_solve.txt"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_noblk.so; do
  echo -n "$lib  " | tee -a "$OUT/ab_lean_block_solve.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$OUT/ab_lean_block_solve.txt"
done; done
for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_noblk.so; do for cfg in "--config C3" "--inputs random"; do
  echo -n "$lib $cfg  " | tee -a "$OUT/ab_lean_block_solve.txt"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us  conv %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['converged_frac_pos_err_lt_1e-4']))" | tee -a "$OUT/ab_lean_block_solve.txt"
done; done
