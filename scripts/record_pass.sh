#!/usr/bin/env bash
# Round-end measurement pass on the MI355X box (run from the repo root through gpurun); everything lands in gpurun_out/rec/.
# Counter passes are separate rocprofv3 runs with --pmc only (no trace domains), as MI355X_MICROARCH.md prescribes.
set -eo pipefail
ROOT="$(pwd)"
OUT="$ROOT/gpurun_out/rec"
mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee "$OUT/pytest_gpu.txt"
echo "== bench (default)"
timeout -k 10 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
cat "$OUT/bench.json"
echo "== bench, one stream"
timeout -k 10 300 python bench.py --no-cpu-baseline --streams 1 > "$OUT/bench_streams1.json" 2>> "$OUT/bench.err"
echo "== bench, random inputs (SURVEY 8d fall-back; worst case for the broad phase)"
timeout -k 10 300 python bench.py --no-cpu-baseline --inputs random > "$OUT/bench_random_inputs.json" 2>> "$OUT/bench.err"
echo "== bench, one-rank RCCL group"
CPPF_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline > "$OUT/bench_dist1.json" 2>> "$OUT/bench.err"
for c in C2 C3 C5; do
    echo "== bench --config $c"
    timeout -k 10 300 python bench.py --no-cpu-baseline --config $c > "$OUT/bench_$c.json" 2>> "$OUT/bench.err"
done
echo "== kbench"
timeout -k 10 600 python scripts/kbench.py > "$OUT/kbench.txt" 2>&1
echo "== launch model"
timeout -k 10 300 python scripts/launch_model.py > "$OUT/launch_model.txt" 2>&1
echo "== iterations per launch (steady state)"
for k in 10 20 30; do timeout -k 10 300 python bench.py --no-cpu-baseline --lm-steps $k | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K =', d['config']['lm_iterations_per_step'], ' step', round(d['ms_per_step']*1e3,2), 'us  isolated kernel', round(d['roofline']['kernel_ms']*1e3,2), 'us')" >> "$OUT/launch_model.txt"; done
echo "== rocprofv3 kernel trace"
cd /tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o kt -- python3 "$ROOT/bench.py" --steps 200 --warmup 10 --no-cpu-baseline --streams 1 > "$OUT/kt_stdout.txt" 2> "$OUT/kt_stderr.txt"
echo "== rocprofv3 pmc FETCH_SIZE"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --streams 1 > /dev/null 2> "$OUT/pmc_fetch_stderr.txt"
echo "== rocprofv3 pmc WRITE_SIZE"
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --prewarm-ms 0 --no-cpu-baseline --streams 1 > /dev/null 2> "$OUT/pmc_write_stderr.txt"
echo "== rocprofv3 pmc SQ instruction counters (two passes)"
timeout -k 10 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d "$OUT/pmc_valu" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_valu_stderr.txt"
timeout -k 10 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d "$OUT/pmc_valu2" -o pmc -- python3 "$ROOT/scripts/pmc_probe.py" panda > /dev/null 2> "$OUT/pmc_valu2_stderr.txt"
cd "$ROOT"
if [ -x build_var/valu_rate ]; then
    echo "== VALU issue-rate calibration (hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rate.hip -o build_var/valu_rate)"
    timeout -k 5 120 ./build_var/valu_rate > "$OUT/valu_issue_rate_calibration.txt" 2>&1
fi
# keep the merge small: drop anything big that is not a csv / json / txt summary
find "$OUT" -type f -size +30M -delete
find "$OUT" -type f | xargs ls -la | awk '{print $5, $9}'
echo "== done"
