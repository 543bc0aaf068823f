#!/usr/bin/env bash
# round 5, GPU call 31: the opt-in fair-share pacing in the product kernel: no result changes, the headline unchanged, one stream paced
set -o pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_lean_parity.py -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/call31_pytest.txt"
for rep in 1 2; do
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['one_stream']; print('driver flags: us/step %.2f  kernel %.2f  one_stream %.2f (kernel %.2f)  paced %.2f (kernel %.2f)  random %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, o['ms_per_step']*1e3, o['kernel_ms']*1e3, o['ms_per_step_paced']*1e3, o['kernel_ms_paced']*1e3, d['random_inputs']['ms_per_step']*1e3))" | tee -a "$OUT/call31_bench.txt"
done
timeout -k 10 600 python bench.py --no-cpu-baseline --no-siblings 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('default flags: us/step %.2f  kernel %.2f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))" | tee -a "$OUT/call31_bench.txt"
