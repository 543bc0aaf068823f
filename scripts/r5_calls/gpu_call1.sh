#!/usr/bin/env bash
# round 5, GPU call 1: the new package-level sharded engine + C-ABI lifetime tests, the lean-iteration parity statistics, the
# profiler's view of the two-launch overlap, a first bench line of the refactored harness.   gpurun -- bash scripts/r5_calls/gpu_call1.sh
set -eo pipefail
ROOT="$(pwd)"; OUT="$ROOT/gpurun_out/r5"; mkdir -p "$OUT"; export TMPDIR=/tmp
echo "== new tests"
timeout -k 10 900 python -m pytest tests/test_gpu_sharded_refiner.py tests/test_gpu_round5.py tests/test_gpu_c_client.py tests/test_gpu_bench_line.py -m gpu -x -q 2>&1 | tail -15 | tee "$OUT/call1_pytest.txt"
echo "== lean-iteration parity test (all cases, no -x)"
timeout -k 10 900 python -m pytest tests/test_gpu_lean_parity.py -m gpu -q -s 2>&1 | tail -60 | tee "$OUT/call1_lean_pytest.txt" || true
echo "== lean-iteration parity statistics"
timeout -k 10 600 python scripts/lean_parity_stats.py > "$OUT/lean_parity.txt" 2> "$OUT/lean_parity.err" || { tail -20 "$OUT/lean_parity.err"; exit 1; }
tail -30 "$OUT/lean_parity.txt"
echo "== rocprofv3 kernel trace of the driver's command (two streams)"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT/overlap" -o kt -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-siblings > "$OUT/overlap_stdout.txt" 2> "$OUT/overlap_stderr.txt"
cd "$ROOT"
python3 scripts/overlap_summary.py "$OUT/overlap" "$OUT/overlap_stdout.txt" "$OUT/r5_overlap.json" 20 | tail -25
rm -rf "$OUT/overlap"
echo "== bench, driver flags"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_driverflags.json" 2> "$OUT/bench.err" || { tail -20 "$OUT/bench.err"; exit 1; }
python3 -c "import json; d=json.load(open('$OUT/bench_driverflags.json')); print('driver flags: us/step %.2f value %.3e kernel %.1f us one_stream %.2f random %.2f cpu agreement %s' % (d['ms_per_step']*1e3, d['value'], d['roofline']['kernel_ms']*1e3, d['one_stream']['ms_per_step']*1e3, d['random_inputs']['ms_per_step']*1e3, d['cpu_baseline']['agreement']))"
echo "== one rank's N = 8 shard, one-rank RCCL, driver flags"
CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --seeds 128 --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench_shard128_driverflags.json" 2>> "$OUT/bench.err"
python3 -c "import json; d=json.load(open('$OUT/bench_shard128_driverflags.json')); print('shard128: default us/step %.2f calibrated %.2f latency_one_request %.2f allgather %.1f us' % (d['ms_per_step']*1e3, d['ms_per_step_calibrated_streams']*1e3, d['latency_one_request']['ms_per_step']*1e3, d['rccl']['allgather_latency_us']))"
