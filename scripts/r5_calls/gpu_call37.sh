#!/usr/bin/env bash
# round 5, GPU call 37: ShardedRefiner(pace=None): tests, then the driver's bench line with the one_stream sibling paced by default
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_sharded_refiner.py tests/test_gpu_bench_line.py -m gpu -x -q 2>&1 | tail -12 | cut -c1-200 | tee "$OUT/call37_pytest.txt"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); o=d['one_stream']; print('driver flags: us/step %.2f  kernel %.2f  one_stream %.2f (kernel %.2f) paced=%s  unpaced %.2f (kernel %.2f)' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3, o['ms_per_step']*1e3, o['kernel_ms']*1e3, o['paced'], o['ms_per_step_unpaced']*1e3, o['kernel_ms_unpaced']*1e3))"
