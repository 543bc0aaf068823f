#!/usr/bin/env bash
# round 5, GPU call 17: the compacted exact capsule tests (queue) -- parity, then A/B against CPPF_COLL_QUEUE=0
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_coll_queue.txt"
timeout -k 10 900 python -m pytest tests/test_gpu_coll_queue.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/call17_pytest.txt"
echo "== A/B: in-tree (queue) against lib_noqueue.so (CPPF_COLL_QUEUE=0)" | tee "$F"
for rep in 1 2; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_noqueue.so; do for cfg in "" "--inputs random" "--config C3"; do
  echo -n "$lib $cfg  " | tee -a "$F"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 300 python bench.py $cfg --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('us/step %.2f   isolated kernel %.2f us  conv %.4f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, d['config']['converged_frac_pos_err_lt_1e-4']))" | tee -a "$F"
done; done; done
