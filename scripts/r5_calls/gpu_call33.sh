#!/usr/bin/env bash
# round 5, GPU call 33: does the (disabled) pacing code cost the headline anything?  A/B against a build with it compiled out
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
F="$OUT/ab_lm_pace_code.txt"; echo "== A/B: in-tree (pacing code in, off) against lib_nopace.so (CPPF_LM_PACE=0: compiled out); 2 000 steps, then the driver's flags" | tee "$F"
timeout -k 10 700 bash scripts/lib_ab.sh build_var/lib_nopace.so 2>&1 | tee -a "$F"
for rep in 1 2 3; do for lib in cppflow_amd/csrc/libcppflow_hip.so build_var/lib_nopace.so; do
  echo -n "$lib  " | tee -a "$F"
  CPPFLOW_HIP_LIB=$lib timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver flags us/step %.2f   isolated kernel %.2f us' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$F"
done; done
